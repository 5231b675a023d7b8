/* Host-side sanitizer probe of the C ABI (tests/test_sanitizers_cpu.py): linked
 * against a build of pita_amd/csrc/ whose HOST code is compiled with the address and undefined-behaviour sanitizers (the device code is
 * not instrumented: GPU sanitizers are unavailable on this pool).  Walks the argument-validation paths of the launch
 * wrappers and the create / destroy paths of every handle type; no kernel is launched, so it runs without a GPU (every
 * valid creation then fails at its first HIP call and must release what it allocated -- LeakSanitizer checks that). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pita_hip.h"
#define EXPECT(cond, code) do { if (!(cond)) { char m[512]; pita_last_error(m, sizeof m); fprintf(stderr, "probe %d failed: rc=%d msg=%s\n", code, rc, m); return code; } } while (0)
int main(void) {
  char msg[64];  /* deliberately short: pita_last_error must truncate, not overrun */
  float* buf = (float*)calloc(64, sizeof(float));
  int rc = 0;
  EXPECT(pita_abi_version() == PITA_ABI_VERSION, 10);
  /* argument validation precedes every HIP call: these run the same on a box without a GPU */
  rc = pita_dw_logp_force(buf, buf, NULL, -1, 4, 2, 1.0f, 0.9f, -4.0f, 0.0f, 4.0f, NULL);
  EXPECT(rc == PITA_EINVAL, 11);
  EXPECT(pita_last_error(msg, sizeof msg) > 0 && strlen(msg) < sizeof msg, 12);
  EXPECT(pita_last_error(msg, 1) >= 0 && msg[0] == 0, 13);
  rc = pita_lj_logp_force(buf, buf, NULL, 4, 13, 3, -1.0f, 1.0f, 1e-6f, 1.0f, 1.0f, 1.0f, NULL);
  EXPECT(rc == PITA_EINVAL, 14);
  rc = pita_lj_logp_force(NULL, buf, NULL, 4, 13, 3, 1.0f, 1.0f, 1e-6f, 1.0f, 1.0f, 1.0f, NULL);
  EXPECT(rc == PITA_EINVAL, 15);
  rc = pita_lj_logp_force(buf, buf, NULL, 0, 13, 3, 1.0f, 1.0f, 1e-6f, 1.0f, 1.0f, 1.0f, NULL);
  EXPECT(rc == PITA_OK, 16);
  rc = pita_lj_mala(buf, buf, NULL, NULL, 8, 7, 3, 1.0f, 1.0f, 1e-6f, 1.0f, 1.0f, 1.0f, 2, NULL, 1, 8, 1, 0, NULL, 0, 1, NULL, NULL, NULL);
  EXPECT(rc == PITA_EUNSUPPORTED, 17);
  rc = pita_lj_mala(buf, buf, NULL, NULL, 8, 13, 3, 1.0f, 1.0f, 1e-6f, 1.0f, 1.0f, 1.0f, 2, NULL, 1, 8, 1, 0, NULL, 0, 1, NULL, NULL, NULL);
  EXPECT(rc == PITA_EINVAL, 18);
  EXPECT(pita_lj_mala_workspace_bytes(-5) == 8 && pita_lj_mala_workspace_bytes(3) == 32, 19);
  rc = pita_quantile_clamp(buf, -3, 1, 0.9f, NULL);
  EXPECT(rc == PITA_EINVAL, 20);
  rc = pita_em_step(buf, buf, NULL, -1, 13, 3, 1e-3f, 1.0f, 0.03f, 1, 0, 0, 1, NULL, NULL);
  EXPECT(rc == PITA_EINVAL, 21);
  rc = pita_systematic_resample(buf, -2, 0.5, NULL, NULL, NULL);
  EXPECT(rc == PITA_EINVAL, 22);
  /* handles: an unsupported configuration never allocates; a short weight vector is refused before it is read */
  pita_egnn_config bad = {13, 3, 64, 3, 2, 1, 1, 15.0f, 0, 1};
  pita_egnn_t* net = NULL;
  rc = pita_egnn_create(&net, &bad, buf, 8);
  EXPECT(rc == PITA_EUNSUPPORTED && net == NULL, 30);
  pita_egnn_config good = {13, 3, 32, 3, 2, 1, 1, 15.0f, 0, 2};
  EXPECT(pita_egnn_num_weights(&good) == 22533, 31);
  rc = pita_egnn_create(&net, &good, buf, 8);
  EXPECT(rc == PITA_EINVAL && net == NULL, 32);
  rc = pita_egnn_create(NULL, &good, buf, 22533);
  EXPECT(rc == PITA_EINVAL, 33);
  /* a complete, valid creation: on a box without a GPU the first HIP call fails and every host allocation made on the
     way must be released (LeakSanitizer); with a GPU the handle is created and destroyed */
  float* w = (float*)calloc(22533, sizeof(float));
  for (int i = 0; i < 22533; ++i) w[i] = 0.01f * (float)((i * 37) % 19 - 9);
  rc = pita_egnn_create(&net, &good, w, 22533);
  if (rc == PITA_OK) { EXPECT(net != NULL, 34); rc = pita_egnn_destroy(net); EXPECT(rc == PITA_OK, 35); }
  else { EXPECT(net == NULL, 36); }
  EXPECT(pita_egnn_destroy(NULL) == PITA_OK || 1, 37);
  pita_egnn_wide_config wc = {22, 3, 64, 5, 7, 1, 1, 1, 15.0f};
  long nw = (long)pita_egnn_wide_num_weights(&wc);
  EXPECT(nw > 0, 40);
  pita_egnn_wide_t* wide = NULL;
  rc = pita_egnn_wide_create(&wide, &wc, buf, 5, buf);
  EXPECT(rc == PITA_EINVAL && wide == NULL, 41);
  float* ww = (float*)calloc((size_t)nw, sizeof(float));
  float* h0 = (float*)calloc(22 * 7, sizeof(float));
  rc = pita_egnn_wide_create(&wide, &wc, ww, nw, h0);
  if (rc == PITA_OK) { rc = pita_egnn_wide_destroy(wide); EXPECT(rc == PITA_OK, 42); } else { EXPECT(wide == NULL, 43); }
  pita_mlp_config mc = {2, 2, 128, 3, 128, 0};
  long nm = (long)pita_mlp_num_weights(&mc);
  EXPECT(nm > 0, 50);
  pita_mlp_t* mlp = NULL;
  float* mw = (float*)calloc((size_t)nm, sizeof(float));
  float fr[64];
  for (int i = 0; i < 64; ++i) fr[i] = 1.0f / (float)(i + 1);
  rc = pita_mlp_create(&mlp, &mc, mw, nm - 1, fr);
  EXPECT(rc == PITA_EINVAL && mlp == NULL, 51);
  rc = pita_mlp_create(&mlp, &mc, mw, nm, fr);
  if (rc == PITA_OK) { rc = pita_mlp_destroy(mlp); EXPECT(rc == PITA_OK, 52); } else { EXPECT(mlp == NULL, 53); }
  int bidx[2] = {0, 1};
  float bpar[2] = {0.1f, 100.0f}, q[2] = {0.1f, -0.1f}, sg[2] = {0.3f, 0.3f}, ep[2] = {0.1f, 0.1f};
  pita_ff_config fc;
  memset(&fc, 0, sizeof fc);
  fc.n_atoms = 2; fc.n_bonds = 1; fc.bond_idx = bidx; fc.bond_par = bpar; fc.charge = q; fc.sigma = sg; fc.epsilon = ep;
  fc.length_scale = 1.0f; fc.kT = 2.5f;
  pita_ff_t* ff = NULL;
  rc = pita_ff_create(&ff, &fc);
  if (rc == PITA_OK) { rc = pita_ff_destroy(ff); EXPECT(rc == PITA_OK, 60); } else { EXPECT(ff == NULL, 61); }
  fc.n_atoms = -1;
  rc = pita_ff_create(&ff, &fc);
  EXPECT(rc == PITA_EINVAL, 62);
  free(buf); free(w); free(ww); free(h0); free(mw);
  printf("sanitized abi ok\n");
  return 0;
}
