// Probe (development aid): what v_cvt_pk_f16_f32 does beyond the f16 range, whether the wave's sticky exception flags
// (TRAPSTS.EXCP) record it, and what the f16 MFMA makes of inf operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(float big, float ok, unsigned* out, float* fout) {
  unsigned mode = __builtin_amdgcn_s_getreg((1 /*HW_REG_MODE*/) | (0 << 6) | (31 << 11));
  __builtin_amdgcn_s_setreg((3 /*HW_REG_TRAPSTS*/) | (0 << 6) | (8 << 11), 0);  // clear EXCP[8:0]
  unsigned t0 = __builtin_amdgcn_s_getreg((3) | (0 << 6) | (31 << 11));
  h2 a = __builtin_convertvector(f2{ok, -ok}, h2);
  unsigned pa = __builtin_bit_cast(unsigned, a);
  asm volatile("" : "+v"(pa));
  unsigned t1 = __builtin_amdgcn_s_getreg((3) | (0 << 6) | (31 << 11));
  h2 b = __builtin_convertvector(f2{big, -big}, h2);
  unsigned pb = __builtin_bit_cast(unsigned, b);
  asm volatile("" : "+v"(pb));
  unsigned t2 = __builtin_amdgcn_s_getreg((3) | (0 << 6) | (31 << 11));
  __builtin_amdgcn_s_setreg((3) | (0 << 6) | (8 << 11), 0);
  float e = __builtin_amdgcn_exp2f(big);  // exp2 overflow
  asm volatile("" : "+v"(e));
  unsigned t3 = __builtin_amdgcn_s_getreg((3) | (0 << 6) | (31 << 11));
  __builtin_amdgcn_s_setreg((3) | (0 << 6) | (8 << 11), 0);
  // MFMA with an inf operand: A = 1 at k = 0, B = {+inf at k=0, -inf at k=1}, A(k=1) = 1
  f16x8 fa = {0}, fb = {0};
  f32x16 acc = {0};
  if (threadIdx.x < 32) { fa[0] = (_Float16)1.0f; fa[1] = (_Float16)1.0f; fb[0] = b.x; fb[1] = b.y; }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  float r0 = acc[0];
  asm volatile("" : "+v"(r0));
  unsigned t4 = __builtin_amdgcn_s_getreg((3) | (0 << 6) | (31 << 11));
  if (threadIdx.x == 0) {
    out[0] = mode; out[1] = t0; out[2] = t1; out[3] = t2; out[4] = t3; out[5] = t4; out[6] = pa; out[7] = pb;
    fout[0] = r0; fout[1] = e;
  }
}

int main() {
  unsigned* d; float* f;
  hipMalloc(&d, 64); hipMalloc(&f, 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 1.0e6f, 3.25f, d, f);
  unsigned h[8]; float hf[2];
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 8, hipMemcpyDeviceToHost);
  printf("MODE = 0x%08x (FP16_OVFL bit 23 = %u)\n", h[0], (h[0] >> 23) & 1);
  printf("TRAPSTS after clear 0x%08x, after in-range cvt 0x%08x, after out-of-range cvt 0x%08x (EXCP bits 8:0; overflow = bit 3)\n", h[1], h[2], h[3]);
  printf("TRAPSTS after exp2(1e6) 0x%08x, after MFMA(inf,-inf) 0x%08x\n", h[4], h[5]);
  printf("cvt_pk_f16_f32(3.25,-3.25) = 0x%08x, cvt_pk_f16_f32(1e6,-1e6) = 0x%08x (0x7c00 = inf, 0x7bff = 65504)\n", h[6], h[7]);
  printf("MFMA 1*(+big16) + 1*(-big16) = %g, exp2(1e6) = %g\n", hf[0], hf[1]);
  return 0;
}
