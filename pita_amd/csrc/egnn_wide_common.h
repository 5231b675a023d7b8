// Native handle behind pita_egnn_wide_t, shared by the two kernels that serve it: the vector-pipe kernel
// (egnn_wide_kernel.hip: any hidden_nf <= 64, any particle count <= 64) and the matrix-pipe kernel
// (egnn_wide_mfma_kernel.hip: the particle systems it is instantiated for).
#pragma once
#include "common.h"

struct pita_egnn_wide {
  pita_egnn_wide_config cfg;
  float* d_w = nullptr;        // vector-pipe kernel: packed weights, see WideLayer
  float* d_estatic = nullptr;  // [n][64] embedding of the static node features + embedding bias, natural feature order
  int device = -1;
  int n_cu = 256;
  // matrix-pipe kernel (null / 0 when the particle system has no instantiation)
  unsigned* d_m16h = nullptr;  // [L][7 matrices][2 x 2 blocks] f16 two-piece fragments
  float* d_vecs64 = nullptr;   // embedding vectors + per-layer vectors, fragment order, f16-path scales folded in
  float* d_est64 = nullptr;    // [n][64] as d_estatic, fragment order
  const void* shape64 = nullptr;
  // fused sampler on the matrix pipe: backup of the walkers + per-particle bookkeeping for the vector-pipe repair pass
  void* d_bk = nullptr;
  size_t bk_bytes = 0;
  int* d_flag = nullptr;  // set by the matrix-pipe kernel when a walker comes out non-finite: the repair pass returns at once otherwise
  int* d_jbad = nullptr;    // [B] flags of the matrix-pipe forward-mode kernel (walkers left to the vector-pipe kernel)
  size_t jbad_bytes = 0;
  float* d_vjp_ws = nullptr;  // reverse-mode kernel (vector pipe): per-wave checkpoints of the forward sweep
  size_t vjp_ws_bytes = 0;
};

namespace pita {
// packs and uploads the matrix-pipe kernel's weights when the particle system has an instantiation (leaves
// net->shape64 null otherwise); w: the flattened state_dict (pita_egnn_wide_create), he: host [n][64] static embedding
int wide64_prepare(pita_egnn_wide* net, const float* w, const float* he);
int wide64_launch(pita_egnn_wide* net, int what, const float* t, const float* x, const float* beta, float* out,
                  long long B, hipStream_t stream);
// n_steps fused Euler-Maruyama steps on the matrix-pipe kernel (mode 3 of egnn_wide64_kernel); bad_from: device [B * n]
int wide64_sampler(pita_egnn_wide* net, float* x, long long B, const float* step_tab, int n_steps, const float* noise,
                   unsigned long long seed, unsigned long long walker_offset, long long step0, int remove_mean,
                   double* stats_out, int* bad_from, hipStream_t stream);
// forward-mode derivative on the matrix pipe (egnn_wide_mfma_jvp_kernel.hip); returns 1 when the particle system has no
// instantiation; bad: device [B] ints, zeroed by the caller, set to 1 for walkers left to the vector-pipe kernel
int wide64_jvp(pita_egnn_wide* net, const float* h, const float* x, const float* beta, const float* vx, int dir,
               const float* vh, float* out, float* dout, float* dot_out, long long dot_stride, long long dot_off,
               float* diag_acc, int* bad, long long B, hipStream_t stream);
void wide64_release(pita_egnn_wide* net);
}  // namespace pita
