// Go / no-go micro-kernel for the walker-resident exact-trace kernel: the MIDDLE layer of the
// tangent sweep with the 39 unit directions as the MFMA column dimension, one workgroup (4 waves, one per SIMD) per
// walker, on synthetic per-edge factors.  Development aid; results quoted in DESIGN.md / profiles/.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/ubench/div_walker_mid tools/ubench/div_walker_mid.hip
//   tools/ubench/div_walker_mid [walkers per block]
//
// Per walker (N = 13 nodes, 156 ordered edges, hidden 32, 39 directions padded to 48 columns = 3 tiles of 16):
//   node phase   Z^B_i = W_b dH_i for the wave's nodes -> f16 two-piece B-operand fragments in LDS (shared by all waves)
//   edge phase   Acc_i += M_ij Z^B_j with M_ij = diag(a) W_2 diag(b) + m' p^T built per edge IN REGISTERS from four
//                32-vectors (16 elements per lane, 3 flops + a 2-instruction f16 split each), 18 v_mfma_f32_16x16x32_f16
//                per edge (2 row blocks x 3 column tiles x 3 products); Abar_i += M_ij beside it
//   node finish  Acc_i += Abar_i (W_a dH_i) + R_i dPos (K = 64), node model dH_i += W_n2 (g_n o (W_n1a dH_i + W_n1b Acc_i))
// Nodes are owned whole by waves (4, 3, 3, 3): the busiest wave bounds the walker.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int N = 13, NT = 3;           // nodes, column tiles of 16
constexpr int ZB_NODE_B = NT * 2 * 1024;  // bytes of one node's B fragments: [tile][piece][lane][16 B]

__device__ __forceinline__ f16x8 as_h8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ float rem_lo(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
__device__ __forceinline__ float rem_hi(unsigned pk, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x));
  return r;
}
// eight fp32 -> two f16 pieces (round to nearest; the remainder is exact in fp32)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2 * q], v[2 * q + 1]}, f16x2));
    const float ra = rem_lo(p1, v[2 * q]), rb = rem_hi(p1, v[2 * q + 1]);
    hi[q] = p1;
    lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, f16x2));
  }
}
struct Frag { u32x4 hi, lo; };  // eight elements of an A or B operand as two f16 pieces
__device__ __forceinline__ f32x4 mma3(const Frag& a, const Frag& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a.lo), as_h8(b.hi), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a.hi), as_h8(b.lo), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a.hi), as_h8(b.hi), c, 0, 0, 0);
  return c;
}
// a 32x32 weight matrix as A operand: [row block 2] fragments, loaded from global (L1/L2 resident)
struct WMat { Frag f[2]; };
__device__ __forceinline__ void load_w(WMat& w, const u32x4* __restrict__ g, int mat, int lane) {
  const u32x4* p = g + (size_t)mat * 256 + lane;
  w.f[0].hi = p[0]; w.f[0].lo = p[64]; w.f[1].hi = p[128]; w.f[1].lo = p[192];
}
// acc[rb][ct] (32 x 48) += W (32x32) x X (32 x 48) with X given as B fragments per column tile
__device__ __forceinline__ void gemm(const WMat& w, const Frag (&x)[NT], f32x4 (&acc)[2][NT]) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) acc[rb][ct] = mma3(w.f[rb], x[ct], acc[rb][ct]);
}
// C layout of a (32 x 48) result = B layout of the next product (k order permuted on the host side of A)
__device__ __forceinline__ void to_frags(const f32x4 (&acc)[2][NT], Frag (&x)[NT]) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const float v[8] = {acc[0][ct].x, acc[0][ct].y, acc[0][ct].z, acc[0][ct].w,
                        acc[1][ct].x, acc[1][ct].y, acc[1][ct].z, acc[1][ct].w};
    split8(v, x[ct].hi, x[ct].lo);
  }
}

struct Params {
  const u32x4* wfrag;   // [6 matrices][2 rb][2 pieces][64 lanes] f16 fragments: Wa Wb W2(unused) Wn1a Wn1b Wn2
  const float* w2f;     // [64 lanes][16] fp32 fragment of W_2 (A layout)
  const float* init;    // random numbers
  float* out;           // [blocks][256][8] sink
  long long* cyc;       // [blocks][4][4] phase cycles
  int walkers;          // per block
};

// node ownership: NW = 4: (3, 3, 3, 4); NW = 8: waves 0-4 two nodes, waves 5-7 one (SIMD s hosts waves s and s + 4)
template <int NW> __device__ __forceinline__ int own_count(int w) { return NW == 4 ? (w == 3 ? 4 : 3) : (w < 5 ? 2 : 1); }
template <int NW> __device__ __forceinline__ int own_first(int w) { return NW == 4 ? 3 * w : (w < 5 ? 2 * w : 10 + (w - 5)); }

template <int NW, int PIPE>
__global__ void __launch_bounds__(NW * 64, NW / 4) mid_kernel(Params p) {
  constexpr int OWN_MAX = NW == 4 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* zb = lds;                                   // [N][ZB_NODE_B]
  unsigned char* dposf = zb + N * ZB_NODE_B;                 // dPos as B fragments, K = 64: [2 ksteps][NT][2][64][16]
  float* fac = reinterpret_cast<float*>(dposf + 2 * ZB_NODE_B);  // [4 waves][12 edges][4][32]
  float* rtab = fac + 4 * 12 * 128;                          // [4 waves][64 lanes][32] R_i in A layout (fp32), 2 rb x 2 ksteps x 8
  float* gnv = rtab + 4 * 64 * 32;                           // [N][32]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
  // synthetic LDS contents
  for (int i = threadIdx.x; i < (N * ZB_NODE_B + 2 * ZB_NODE_B) / 4; i += NW * 64)
    reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + ((i * 2654435761u) & 0x03ff03ffu);  // f16 pairs in [1, 2)
  for (int i = threadIdx.x; i < 4 * 12 * 128 + 4 * 64 * 32 + N * 32; i += NW * 64) fac[i] = 0.25f + 0.5f * p.init[i & 4095];
  __syncthreads();

  float w2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) w2[q] = p.w2f[lane * 16 + q];
  // tangent features of the owned nodes: dH[n][rb][ct]
  f32x4 dH[OWN_MAX][2][NT];
#pragma unroll
  for (int n = 0; n < OWN_MAX; ++n)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
        for (int q = 0; q < 4; ++q) dH[n][rb][ct][q] = p.init[(threadIdx.x * 97 + n * 24 + rb * 12 + ct * 4 + q) & 4095] - 0.5f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  long long tA = 0, tB = 0, tC = 0;
  const int nown = own_count<NW>(wave), first = own_first<NW>(wave);
  const float* myfac = fac + (wave & 3) * 12 * 128;
  const float* myr = rtab + (wave & 3) * 64 * 32 + lane * 32;

  for (int it = 0; it < p.walkers; ++it) {
    long long t0 = __builtin_amdgcn_s_memtime();
    // ---- node phase: Z^B_i = W_b dH_i -> LDS fragments
    {
      WMat wb;
      load_w(wb, p.wfrag, 1, lane);
#pragma unroll
      for (int n = 0; n < OWN_MAX; ++n) {
        if (n >= nown) continue;
        Frag x[NT];
        to_frags(dH[n], x);
        f32x4 z[2][NT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) z[rb][ct] = zero4;
        gemm(wb, x, z);
        Frag o[NT];
        to_frags(z, o);
        u32x4* dst = reinterpret_cast<u32x4*>(zb + (first + n) * ZB_NODE_B) + lane;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { dst[ct * 128] = o[ct].hi; dst[ct * 128 + 64] = o[ct].lo; }
      }
    }
    __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    // ---- edge phase + node finish, node by node
#pragma unroll
    for (int n = 0; n < OWN_MAX; ++n) {
      if (n >= nown) continue;
      const int i = first + n;
      f32x4 acc[2][NT];
      float abar[16], abar3[8];
      f32x4 acc3[NT];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[rb][ct] = zero4;
#pragma unroll
      for (int q = 0; q < 16; ++q) abar[q] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) abar3[q] = 0.f;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc3[ct] = zero4;
      // third row block: rows 32..34 = c_k q^T (the coordinate head's adjoint), lanes r16 >= 3 carry zeros
      auto build = [&](int jj, Frag (&A)[3]) {
        const float* f = myfac + jj * 128;
        const f32x2 am0 = *reinterpret_cast<const f32x2*>(f + 2 * r16);
        const f32x2 am1 = *reinterpret_cast<const f32x2*>(f + 32 + 2 * r16);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(f + 64 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(f + 64 + 8 * g + 4);
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(f + 96 + 8 * g), p1 = *reinterpret_cast<const f32x4*>(f + 96 + 8 * g + 4);
        const float bk[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float pk[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const float a = rb ? am1.x : am0.x, mp = rb ? am1.y : am0.y;
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            v[q] = fmaf(a, w2[rb * 8 + q], mp * pk[q]) * bk[q];
            abar[rb * 8 + q] += v[q];
          }
          split8(v, A[rb].hi, A[rb].lo);
        }
        {
          const float c = myfac[jj * 128 + (r16 & 3)] * (r16 < 3 ? 1.0f : 0.0f);
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            v[q] = c * pk[q];
            abar3[q] += v[q];
          }
          split8(v, A[2].hi, A[2].lo);
        }
      };
      auto mma = [&](const Frag (&A)[3], int j) {
        const u32x4* src = reinterpret_cast<const u32x4*>(zb + j * ZB_NODE_B) + lane;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          Frag B;
          B.hi = src[ct * 128];
          B.lo = src[ct * 128 + 64];
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc[rb][ct] = mma3(A[rb], B, acc[rb][ct]);
          acc3[ct] = mma3(A[2], B, acc3[ct]);
        }
      };
      if (PIPE == 0) {
        for (int jj = 0; jj < N - 1; ++jj) {
          Frag A[3];
          build(jj, A);
          mma(A, jj + (jj >= i));
        }
      } else {
        Frag Ac[3];
        build(0, Ac);
        for (int jj = 0; jj < N - 1; ++jj) {
          Frag An[3];
          build(jj + 1 < N - 1 ? jj + 1 : jj, An);  // (the last iteration rebuilds its own edge: same work, discarded)
          mma(Ac, jj + (jj >= i));
#pragma unroll
          for (int q = 0; q < 27; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, PIPE, 0);
          }
#pragma unroll
          for (int rb = 0; rb < 3; ++rb) Ac[rb] = An[rb];
        }
      }
      // ---- node finish
      Frag xh[NT];
      to_frags(dH[n], xh);
      {  // Acc += Abar (W_a dH_i)
        WMat wa;
        load_w(wa, p.wfrag, 0, lane);
        f32x4 za[2][NT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) za[rb][ct] = zero4;
        gemm(wa, xh, za);
        Frag xa[NT];
        to_frags(za, xa);
        WMat ab;
        const float v0[8] = {abar[0], abar[1], abar[2], abar[3], abar[4], abar[5], abar[6], abar[7]};
        const float v1[8] = {abar[8], abar[9], abar[10], abar[11], abar[12], abar[13], abar[14], abar[15]};
        split8(v0, ab.f[0].hi, ab.f[0].lo);
        split8(v1, ab.f[1].hi, ab.f[1].lo);
        gemm(ab, xa, acc);
      }
      {  // Acc += R_i dPos, K = 64
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          WMat rm;
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(myr + (ks * 2 + rb) * 8), a1 = *reinterpret_cast<const f32x4*>(myr + (ks * 2 + rb) * 8 + 4);
            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            split8(v, rm.f[rb].hi, rm.f[rb].lo);
          }
          Frag dp[NT];
          const u32x4* src = reinterpret_cast<const u32x4*>(dposf + ks * ZB_NODE_B) + lane;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) { dp[ct].hi = src[ct * 128]; dp[ct].lo = src[ct * 128 + 64]; }
          gemm(rm, dp, acc);
        }
      }
      {  // node model
        WMat wn;
        f32x4 zn[2][NT];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) zn[rb][ct] = zero4;
        load_w(wn, p.wfrag, 3, lane);
        gemm(wn, xh, zn);
        Frag xg[NT];
        to_frags(acc, xg);
        load_w(wn, p.wfrag, 4, lane);
        gemm(wn, xg, zn);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gnv + i * 32 + 4 * g), g1 = *reinterpret_cast<const f32x4*>(gnv + i * 32 + 16 + 4 * g);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) { zn[0][ct] *= g0; zn[1][ct] *= g1; }
        Frag xz[NT];
        to_frags(zn, xz);
        load_w(wn, p.wfrag, 5, lane);
        gemm(wn, xz, dH[n]);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) dH[n][0][ct] += acc3[ct] * abar3[ct];
        // keep the synthetic state bounded (stands in for nothing: a real sweep's tangents are not renormalised)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) dH[n][rb][ct] *= 1.0f / 64.0f;
      }
    }
    long long t2 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    long long t3 = __builtin_amdgcn_s_memtime();
    tA += t1 - t0; tB += t2 - t1; tC += t3 - t2;
  }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < OWN_MAX; ++n)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) s += dH[n][rb][ct].x + dH[n][rb][ct].y + dH[n][rb][ct].z + dH[n][rb][ct].w;
  p.out[blockIdx.x * NW * 64 + threadIdx.x] = s;
  if (lane == 0) {
    long long* c = p.cyc + (blockIdx.x * 8 + wave) * 4;
    c[0] = tA; c[1] = tB; c[2] = tC;
  }
}

int main(int argc, char** argv) {
  const int walkers = argc > 1 ? atoi(argv[1]) : 64;
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int blocks = prop.multiProcessorCount;
  std::vector<float> init(4096), w2(64 * 16);
  std::vector<unsigned> wf(6 * 256 * 4);
  srand(1);
  for (auto& v : init) v = (float)(rand() & 0xffffff) / 16777216.0f;
  for (auto& v : w2) v = ((float)(rand() & 0xffffff) / 16777216.0f - 0.5f) * 0.35f;
  for (size_t i = 0; i < wf.size(); ++i) {  // hi pieces ~ +-0.2, lo pieces ~ 1e-4 (f16 bit patterns)
    const bool lo = ((i / 4 / 64) & 1) != 0;
    const unsigned short a = (unsigned short)((lo ? 0x0a00 : 0x3000) + (rand() & 0x3ff) + ((rand() & 1) << 15));
    const unsigned short b = (unsigned short)((lo ? 0x0a00 : 0x3000) + (rand() & 0x3ff) + ((rand() & 1) << 15));
    wf[i] = a | ((unsigned)b << 16);
  }
  float *d_init, *d_w2, *d_out;
  unsigned* d_wf;
  long long* d_cyc;
  CHECK(hipMalloc(&d_init, init.size() * 4));
  CHECK(hipMalloc(&d_w2, w2.size() * 4));
  CHECK(hipMalloc(&d_wf, wf.size() * 4));
  CHECK(hipMalloc(&d_out, blocks * 512 * 4));
  CHECK(hipMalloc(&d_cyc, blocks * 32 * 8));
  CHECK(hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_w2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_wf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
  Params p{reinterpret_cast<const u32x4*>(d_wf), d_w2, d_init, d_out, d_cyc, walkers};
  const size_t lds = N * ZB_NODE_B + 2 * ZB_NODE_B + 4 * (4 * 12 * 128 + 4 * 64 * 32 + N * 32);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel<8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 40; ++rep) {
    const int var = rep % 5;
    const int NW = var == 1 ? 8 : 4;
    CHECK(hipEventRecord(e0));
    if (var == 0) hipLaunchKernelGGL((mid_kernel<4, 0>), dim3(blocks), dim3(256), lds, 0, p);
    else if (var == 1) hipLaunchKernelGGL((mid_kernel<8, 0>), dim3(blocks), dim3(512), lds, 0, p);
    else if (var == 2) hipLaunchKernelGGL((mid_kernel<4, 2>), dim3(blocks), dim3(256), lds, 0, p);
    else if (var == 3) hipLaunchKernelGGL((mid_kernel<4, 3>), dim3(blocks), dim3(256), lds, 0, p);
    else hipLaunchKernelGGL((mid_kernel<4, 4>), dim3(blocks), dim3(256), lds, 0, p);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    {  // run-to-run determinism of the whole output
      static std::vector<float> first[5];
      std::vector<float> cur((size_t)blocks * NW * 64);
      CHECK(hipMemcpy(cur.data(), d_out, cur.size() * 4, hipMemcpyDeviceToHost));
      if (first[var].empty()) first[var] = cur;
      else {
        size_t nd = 0;
        for (size_t i = 0; i < cur.size(); ++i) nd += (memcmp(&cur[i], &first[var][i], 4) != 0);
        printf("variant %d: %zu of %zu outputs differ from the first launch of this variant\n", var, nd, cur.size());
      }
    }
    std::vector<long long> cyc(blocks * 32);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double a[8] = {0}, b[8] = {0}, c[8] = {0};
    for (int bl = 0; bl < blocks; ++bl)
      for (int w = 0; w < NW; ++w) { a[w] += cyc[(bl * 8 + w) * 4]; b[w] += cyc[(bl * 8 + w) * 4 + 1]; c[w] += cyc[(bl * 8 + w) * 4 + 2]; }
    const double per = 1.0 / ((double)blocks * walkers);
    printf("rep %d (%d waves, variant %d): %.3f ms for %d walkers per CU x %d CUs = %.2f us per walker and CU  (LDS %zu B)\n", rep, NW, var, ms, walkers, blocks,
           1e3 * ms / walkers, lds);
    for (int w = 0; w < NW; ++w)
      printf("   wave %d: node phase %.0f, edge + finish %.0f, wait at barrier %.0f cycles per walker\n", w, a[w] * per, b[w] * per, c[w] * per);
    printf("   => 65 536 walkers on %d CUs: %.2f ms for this layer\n", blocks, 65536.0 / blocks * (ms / walkers));
  }
  float sink;
  CHECK(hipMemcpy(&sink, d_out, 4, hipMemcpyDeviceToHost));
  printf("sink %g\n", sink);
  return 0;
}
