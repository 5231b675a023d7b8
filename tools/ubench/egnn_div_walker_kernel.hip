// RECORDED NEGATIVE (round 5; moved out of the product library in round 6).  This file is kept as documentation of the
// walker-resident mapping and of its algebra; it is NOT built into libpita_hip.so any more.  It was wired into the library
// through three hooks (commit e5f94e2 has them): pita::wk_pack_create() from pita_egnn_create (egnn_kernel.hip), the
// wk_available()/wk_launch() branch at the top of pita_egnn_jacobian_trace (egnn_div_kernel.hip), and pita_egnn::d_wk
// (egnn_common.h), behind PITA_DIV_WALKER=1, compiled with -Xclang -target-feature -Xclang -packed-fp32-ops.  Measured:
// 30.4 ms per 65 536-walker LJ13 trace against 19.0 for the cached path (profiles/r05_walker_*.txt).
//
// Walker-resident exact trace of the EDM-preconditioned EGNN denoiser's Jacobian for gfx950: ONE launch for all N dim unit
// directions, the directions are the COLUMN dimension of the matrix instructions, nothing is cached in HBM.
//
// What it replaces: the reference computes div_x s_theta = (trace(J_x D) - dim) / h with vmap(jacrev)
// (pita/src/models/components/utils.py:30-51, called from sdes.py:151-239) through EGNN_dynamics
// (egnn_temp_conditioned.py:56-93, E_GCL :197-356).  egnn_div_kernel.hip keeps the forward kernel's mapping (column =
// (walker, node)), replicates registers per direction and streams a 180 KB-per-walker primal cache three times per step.
// Here a workgroup owns ONE walker and the algebra is turned round (tools/div_walker_emulate.py checks it in fp64 against
// autograd): per edge (i, j) the whole map from the first edge layer's pre-activation tangent dz1 to the message tangent
// AND to the coordinate head's contribution to the position tangent is ONE (32 + dim) x 32 matrix
//     M_ij = [ diag(att g2) W2 diag(g1) + m' p^T ;  (Dhat_k tau) q^T  (k < dim) ]
// built in registers from five 32-vectors of the primal (a = att g2, b = g1, m' = att (1 - att) m, p = g1 o W2^T (w_att o g2),
// q = M^T v_c), and dz1_ij[:, d] = Wa dH_i[:, d] + Wb dH_j[:, d] + w_r dr_ij[d] + w_e de_ij[d] for all directions d at once:
//     Acc_i[(32 + dim) x D]  =  sum_j M_ij (Wb dH_j)  +  (sum_j M_ij) (Wa dH_i)  +  [alpha_ij | eps_ij | phi/(|d|+1)] S_i
// -- the first term 27 v_mfma_f32_16x16x32_f16 per edge against B fragments of Wb dH_j that every wave reads from LDS (f16
// two-piece operands, three products: fp32-equivalent), the second one product per node, the third one K = 64 product per
// node whose B operand S_i = [dr_ij | d Delta_ij | de_ij] is formed on the fly from the position tangents.  Rows 0..31 of
// Acc_i are d agg_i (node model), rows 32.. the update of d pos_i.  The first layer (dH = 0) has no edge product, the last
// one only the coordinate rows.
//
// Mapping: 8 waves per workgroup (two per SIMD: one wave's operand construction issues beside its partner's matrix
// instructions), node i belongs to wave i mod 8; per layer two workgroup barriers.  16 x 16 x 32 tiles: lane (c, g) = (l & 15,
// l >> 4) holds column c and the eight features F(g) = {4g..4g+3, 16+4g..16+4g+3} of a result, which is at once the B operand
// fragment of the next product when that product's A operand is packed with its k index in "slot order"
// (slot 8g + e <-> feature F(g)[e], wk_feat): layers chain without data movement.
//
// f16 range: weights x 16, feature tangents x 32, the per-edge matrix x 64, coordinate rows by a per-node power of two
// taken from the tile's largest entry (the coordinate head of a fresh net is ~1e-4, of a trained one ~1e-1); a walker whose
// result is not finite is marked and recomputed by egnn_div_kernel (bf16 three-piece) exactly as the other fast paths do.
#include "egnn_common.h"

namespace pita {

constexpr int WK_NW = 8;  // waves per workgroup
enum { WM_WA = 0, WM_WB, WM_W2, WM_WC1, WM_WN1A, WM_WN1B, WM_WN2, WM_WC1T, WM_W2T, WM_COUNT };
constexpr int WK_MAT_W = 1024;  // 32-bit words per matrix: [row block 2][piece 2][lane 64][4]
enum { WV_WR = 0, WV_WE, WV_B1, WV_B2, WV_WATT, WV_BC1, WV_WC2, WV_BN1, WV_BN2, WV_COUNT };  // 32-vectors in slot order
constexpr int WK_VEC_LAYER_F = WV_COUNT * 32 + 4;  // + b_att (+ pad)
constexpr int WK_VEC_EMB_F = 96;                   // emb_w0, emb_w1, emb_b in slot order
constexpr float WK_SW = 16.0f;    // weight fragments
constexpr float WK_SA = 64.0f;    // per-edge matrix, rows 0..31
constexpr float WK_ST = 32.0f;    // feature tangents dH
constexpr float WK_SS = 512.0f;   // S_i (dr, d Delta, de)
constexpr float WK_SZ = WK_SW * WK_ST;  // Wa dH, Wb dH as they stand in the accumulators (512)

__host__ __device__ constexpr int wk_feat(int slot) {
  return (slot & 7) < 4 ? 4 * (slot >> 3) + (slot & 7) : 16 + 4 * (slot >> 3) + ((slot & 7) - 4);
}

struct WkParams {
  const unsigned* mats;  // [L][WM_COUNT][WK_MAT_W]
  const float* w2f;      // [L][64 lanes][16]  WK_SA W2 as an fp32 A fragment (slot-ordered k)
  const float* vecs;     // [WK_VEC_EMB_F + L WK_VEC_LAYER_F]
  int n_layers, in_nf, attention, tanh_on, feature_layout;
  float coord_scale;
  long long B;
  const float* h;
  const float* x;
  const float* beta;
  float* trace;  // [B] += trace(J_x D)
  float* out;    // [B, D] denoiser or null
  int* mark;
  int* bad_flag;
  int bad_seq;
};

// development aid (-DPITA_WK_STAMPS): shader cycles per section of the kernel, summed over all waves
#ifdef PITA_WK_STAMPS
__device__ unsigned long long wk_dbg[16];
#define WK_STAMP(k)                                                        \
  do {                                                                     \
    const long long _t = __builtin_amdgcn_s_memtime();                     \
    wk_acc[k] += (unsigned long long)(_t - wk_t0);                         \
    wk_t0 = _t;                                                            \
  } while (0)
#else
#define WK_STAMP(k) do { } while (0)
#endif


struct WkFrag { u32x4 hi, lo; };  // eight operand elements as two f16 pieces
struct WkMat { WkFrag f[2]; };    // a 32 x 32 A operand (two row blocks)

__device__ __forceinline__ void wk_split8(const float (&v)[8], WkFrag& o) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v[2 * q], v[2 * q + 1]}, f16x2));
    const float ra = f16_rem_lo(p1, v[2 * q]), rb = f16_rem_hi(p1, v[2 * q + 1]);
    o.hi[q] = p1;
    o.lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{ra, rb}, f16x2));
  }
}
__device__ __forceinline__ f32x4 wk_mma3(const WkFrag& a, const WkFrag& b, f32x4 c) {  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a.lo), as_f16x8(b.hi), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a.hi), as_f16x8(b.lo), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a.hi), as_f16x8(b.hi), c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ void wk_load(WkMat& w, const unsigned* __restrict__ mats, int mat, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(mats + (size_t)mat * WK_MAT_W) + lane;
  w.f[0].hi = p[0]; w.f[0].lo = p[64]; w.f[1].hi = p[128]; w.f[1].lo = p[192];
}
template <int NT>
__device__ __forceinline__ void wk_gemm(const WkMat& w, const WkFrag (&x)[NT], f32x4 (&acc)[2][NT]) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) acc[rb][ct] = wk_mma3(w.f[rb], x[ct], acc[rb][ct]);
}
// the eight values a lane holds of a 32-row result (slot order) -> the B fragment of the next product
template <int NT>
__device__ __forceinline__ void wk_frags(const f32x4 (&acc)[2][NT], float scale, WkFrag (&x)[NT]) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const float v[8] = {acc[0][ct].x * scale, acc[0][ct].y * scale, acc[0][ct].z * scale, acc[0][ct].w * scale,
                        acc[1][ct].x * scale, acc[1][ct].y * scale, acc[1][ct].z * scale, acc[1][ct].w * scale};
    wk_split8(v, x[ct]);
  }
}
__device__ __forceinline__ void wk_silu_d(float z, float& y, float& g) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
  y = z * s;
  g = s * fmaf(z, 1.0f - s, 1.0f);
}
// sum over the four lanes (c, g = 0..3) that share a column: the 32 features of one dot product
__device__ __forceinline__ float wk_gsum(float v) {
  // v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows / halves of two registers inside the VALU: swapping two
  // copies of v leaves (even rows, even rows) and (odd rows, odd rows): their sum is v + v[lane ^ 16]; likewise for 32
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  return xhalf_sum(v);
}
// sum over the 16 columns of a tile (the lanes of one DPP row); every lane of the row gets the sum
__device__ __forceinline__ float wk_rowsum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}
__device__ __forceinline__ float wk_wavemax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void wk_ld8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void wk_st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ float wk_dot8(const float (&a)[8], const float (&b)[8]) {
  float s = a[0] * b[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) s = fmaf(a[e], b[e], s);
  return s;
}

template <int N, int DIM>
struct WkCfg {
  static constexpr int D = N * DIM;
  static constexpr int NT = (D + 15) / 16;        // column tiles
  static constexpr int NP = N - 1;                // partners of a node
  static constexpr int OWN = (N + WK_NW - 1) / WK_NW;
  static_assert(NP <= 12 && DIM <= 3 && NT <= 3, "S-product slot layout: 4 slots per partner in 48, de in the last 16");
  static constexpr int ZB_NODE = NT * 2 * 1024;   // bytes of a full set of B fragments [tile][piece][lane][16 B]
  // Wb dH_j fragments as every wave reads them: the last column tile has D - 16 (NT - 1) live columns; when that is <= 8 it is
  // stored for 8 columns only ([piece][g][c & 7][16 B]: lanes c >= 8 read the columns c - 8 again, results nobody uses)
  static constexpr bool ZB_HALF = (D - 16 * (NT - 1)) <= 8;
  static constexpr int ZB_LAST_PIECE = ZB_HALF ? 512 : 1024;
  static constexpr int ZB_NODE_C = (NT - 1) * 2048 + 2 * ZB_LAST_PIECE;
  static __device__ __forceinline__ int zb_off(int ct, int piece, int c, int g) {
    return (ct == NT - 1 && ZB_HALF) ? ct * 2048 + piece * 512 + ((c & 7) + 8 * g) * 16 : ct * 2048 + piece * 1024 + (c + 16 * g) * 16;
  }
  // per-edge record of the factor table (floats)
  static constexpr int FT_AM = 0, FT_B = 64, FT_P = 96, FT_Q = 128, FT_C3 = 160, FT_X = 164, FT_STRIDE = 176;
  static constexpr int FT_F = NP * FT_STRIDE;
  // tables overlaid on the factor table once the edge loop has read it: R_i [35 rows][K slot 4 j + k] (row stride 60 floats:
  // conflict-free ds_read_b128 of 16 rows), then E_i^T [D directions][36 rows]
  static constexpr int R_STRIDE = 60, E_STRIDE = 36;
  static_assert(4 * N <= 56, "position tangents: four K slots per node in 64");
  static_assert(35 * R_STRIDE <= FT_F && D * E_STRIDE <= FT_F, "the position-term tables fit the factor table");
  // LDS carve-up (bytes)
  static constexpr int O_ZB = 0;
  static constexpr int O_DPOS = O_ZB + N * ZB_NODE_C;            // WK_SS x d pos entering the layer as B fragments, K = 64: [kstep 2][ZB_NODE];
                                                               // slot 4 j + k = (node j, coordinate k): a node's slots are whole
                                                               // dwords of ONE owner wave (no sub-dword stores from two waves)
  static constexpr int O_POS = O_DPOS + 2 * ZB_NODE;           // float4 [N] pos, float4 [N] pos0
  static constexpr int O_ZA = O_POS + 2 * N * 16;              // float [N][32] Wa h + b1 (slot order), then [N][32] Wb h
  static constexpr int O_VEC = O_ZA + 2 * N * 32 * 4;          // float [WK_VEC_LAYER_F]
  static constexpr int O_DIAG = O_VEC + ((WK_VEC_LAYER_F * 4 + 15) / 16) * 16;  // float [N * DIM (pad 48)] + scratch
  static constexpr int O_NODE = O_DIAG + 64 * 4;               // float [WK_NW][48]: a node's aggregate and position update
  static constexpr int O_DPN = O_NODE + WK_NW * 48 * 4;        // float4 [N][NT * 16]: d pos of every node (true scale), touched by its owner only
  static constexpr int O_PNEW = O_DPN + N * NT * 16 * 16;      // float4 [N]: positions leaving the layer
  static constexpr int O_FT = O_PNEW + N * 16;
  static constexpr int LDS_BYTES = O_FT + WK_NW * FT_F * 4;
};

template <int N, int DIM>
__global__ void __launch_bounds__(WK_NW * 64, 2) egnn_div_walker_kernel(WkParams p) {
  using C = WkCfg<N, DIM>;
  constexpr int NT = C::NT, NP = C::NP, D = C::D;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* zb = lds + C::O_ZB;
  unsigned char* dposF = lds + C::O_DPOS;
  f32x4* posT = reinterpret_cast<f32x4*>(lds + C::O_POS);
  float* zaT = reinterpret_cast<float*>(lds + C::O_ZA);
  float* zbT = zaT + N * 32;
  float* vecL = reinterpret_cast<float*>(lds + C::O_VEC);
  float* diag = reinterpret_cast<float*>(lds + C::O_DIAG);
  const int wave = threadIdx.x >> 6;
  int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  // re-derive the lane coordinates from an opaque copy at the head of every phase: left alone the compiler hoists every
  // lane-dependent LDS address of the whole walker loop to the top of the kernel and spills them (64 of its scratch slots)
#define WK_RELANE()                               \
  do {                                            \
    int l_ = threadIdx.x & 63;                    \
    asm volatile("" : "+v"(l_));                  \
    lane = l_; c = l_ & 15; g = l_ >> 4;          \
  } while (0)
  float* FT = reinterpret_cast<float*>(lds + C::O_FT) + wave * C::FT_F;
  float* nodeS = reinterpret_cast<float*>(lds + C::O_NODE) + wave * 48;
  f32x4* dpn = reinterpret_cast<f32x4*>(lds + C::O_DPN);
  f32x4* posNew = reinterpret_cast<f32x4*>(lds + C::O_PNEW);
  const int nown = (wave + WK_NW < N) ? 2 : (wave < N ? 1 : 0);
  const int L = p.n_layers;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* vemb = p.vecs;

  for (int t = threadIdx.x; t < 2 * C::ZB_NODE / 16; t += WK_NW * 64) reinterpret_cast<u32x4*>(dposF)[t] = u32x4{0u, 0u, 0u, 0u};
  // d pos of node i (lanes g = 0: rows k of column 16 ct + c) -> its K slots 3 i + k of the shared B fragments
  auto publish_dpos = [&](int i) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    if (g == 0) {
      const int t = 4 * i;  // elements t & 7 = 0 or 4 of lane group (t & 31) >> 3: one 8-byte store per tile and piece
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const f32x4 dp = dpn[i * (NT * 16) + 16 * ct + c];
        const float v0 = WK_SS * dp.x, v1 = WK_SS * dp.y, v2 = WK_SS * dp.z;
        const f16x2 h01 = __builtin_convertvector(f32x2_t{v0, v1}, f16x2), h2 = __builtin_convertvector(f32x2_t{v2, 0.f}, f16x2);
        const f16x2 l01 = __builtin_convertvector(f32x2_t{v0 - (float)h01.x, v1 - (float)h01.y}, f16x2);
        const f16x2 l2 = __builtin_convertvector(f32x2_t{v2 - (float)h2.x, 0.f}, f16x2);
        unsigned char* base = dposF + (t >> 5) * C::ZB_NODE + (c + 16 * ((t & 31) >> 3)) * 16 + (t & 7) * 2;
        *reinterpret_cast<u32x2_t*>(base + (ct * 2) * 1024) = u32x2_t{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h2)};
        *reinterpret_cast<u32x2_t*>(base + (ct * 2 + 1) * 1024) = u32x2_t{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l2)};
      }
    }
  };
#ifdef PITA_WK_STAMPS
  long long wk_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long wk_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (long long wid = blockIdx.x; wid < p.B; wid += gridDim.x) {
    WK_STAMP(0);
    // ---------------------------------------------------------------- walker set-up
    WK_RELANE();
    const float hv = p.h[wid], bet = p.beta ? p.beta[wid] : 0.f;
    const float op = 1.0f + hv, rs = 1.0f / sqrtf(op);
    const float c_s = 1.0f / op, c_in = rs, c_out = sqrtf(hv) * rs, tfeat = 0.125f * logf(hv);
    __syncthreads();  // the previous walker's epilogue has read posT / diag
    if (threadIdx.x < N) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < DIM; ++k) v[k] = c_in * p.x[(wid * N + threadIdx.x) * DIM + k];
      posT[threadIdx.x] = f32x4{v[0], v[1], v[2], 0.f};
      posT[N + threadIdx.x] = f32x4{v[0], v[1], v[2], 0.f};
    }
    // node features of the owned nodes, column c = own slot (quirk Q1: egnn_temp_conditioned.py:68-78)
    float hq[8];
    {
      const int node = wave + WK_NW * c;
      float a0 = tfeat, a1 = 0.f;
      if (p.in_nf == 2) {
        if (p.feature_layout == 0) { a0 = (2 * node < N) ? tfeat : bet; a1 = (2 * node + 1 < N) ? tfeat : bet; }
        else a1 = bet;
      }
      float w0[8], w1[8], eb[8];
      wk_ld8(vemb + 8 * g, w0); wk_ld8(vemb + 32 + 8 * g, w1); wk_ld8(vemb + 64 + 8 * g, eb);
#pragma unroll
      for (int e = 0; e < 8; ++e) hq[e] = (c < nown) ? fmaf(w0[e], a0, fmaf(w1[e], a1, eb[e])) : 0.f;
    }
    f32x4 dHs[C::OWN][2][NT];  // WK_ST x the feature tangents of the owned nodes
#pragma unroll
    for (int s = 0; s < C::OWN; ++s) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        dHs[s][0][ct] = zero4; dHs[s][1][ct] = zero4;
        const int rel = 16 * ct + c - (wave + WK_NW * s) * DIM;  // unit directions: d pos_i[k][d] = [d == i dim + k]
        if (s < nown && g == 0)
          dpn[(wave + WK_NW * s) * (NT * 16) + 16 * ct + c] = f32x4{rel == 0 ? 1.f : 0.f, rel == 1 ? 1.f : 0.f, (DIM > 2 && rel == 2) ? 1.f : 0.f, 0.f};
      }
      if (s < nown) publish_dpos(wave + WK_NW * s);
    }

    WK_STAMP(1);
    for (int l = 0; l < L; ++l) {
      const bool first = l == 0, last = l == L - 1;
      const unsigned* mats = p.mats + (size_t)l * WM_COUNT * WK_MAT_W;
      const float* vl = p.vecs + WK_VEC_EMB_F + (size_t)l * WK_VEC_LAYER_F;
      // ------------------------------------------------------------ layer start: tables every wave reads
      WK_RELANE();
      if (threadIdx.x < WK_VEC_LAYER_F) vecL[threadIdx.x] = vl[threadIdx.x];
      {  // Wa h + b1, Wb h of the owned nodes (columns 0, 1)
        WkFrag xh[1];
        wk_split8(hq, xh[0]);
        WkMat wm;
        f32x4 z[2][1];
        float b1[8];
        wk_ld8(vl + WV_B1 * 32 + 8 * g, b1);
        wk_load(wm, mats, WM_WA, lane);
        z[0][0] = zero4; z[1][0] = zero4;
        wk_gemm<1>(wm, xh, z);
        if (c < nown) {
          const float v[8] = {fmaf(z[0][0].x, 1.f / WK_SW, b1[0]), fmaf(z[0][0].y, 1.f / WK_SW, b1[1]), fmaf(z[0][0].z, 1.f / WK_SW, b1[2]),
                              fmaf(z[0][0].w, 1.f / WK_SW, b1[3]), fmaf(z[1][0].x, 1.f / WK_SW, b1[4]), fmaf(z[1][0].y, 1.f / WK_SW, b1[5]),
                              fmaf(z[1][0].z, 1.f / WK_SW, b1[6]), fmaf(z[1][0].w, 1.f / WK_SW, b1[7])};
          wk_st8(zaT + (wave + WK_NW * c) * 32 + 8 * g, v);
        }
        wk_load(wm, mats, WM_WB, lane);
        z[0][0] = zero4; z[1][0] = zero4;
        wk_gemm<1>(wm, xh, z);
        if (c < nown) {
          const float v[8] = {z[0][0].x * (1.f / WK_SW), z[0][0].y * (1.f / WK_SW), z[0][0].z * (1.f / WK_SW), z[0][0].w * (1.f / WK_SW),
                              z[1][0].x * (1.f / WK_SW), z[1][0].y * (1.f / WK_SW), z[1][0].z * (1.f / WK_SW), z[1][0].w * (1.f / WK_SW)};
          wk_st8(zbT + (wave + WK_NW * c) * 32 + 8 * g, v);
        }
        if (!first) {  // Wb dH_i as B fragments (x WK_SZ)
#pragma unroll
          for (int s = 0; s < C::OWN; ++s) {
            if (s >= nown) continue;
            WkFrag xd[NT];
            wk_frags<NT>(dHs[s], 1.0f, xd);
            f32x4 zz[2][NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) { zz[0][ct] = zero4; zz[1][ct] = zero4; }
            wk_gemm<NT>(wm, xd, zz);
            WkFrag o[NT];
            wk_frags<NT>(zz, 1.0f, o);
            unsigned char* dst = zb + (wave + WK_NW * s) * C::ZB_NODE_C;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
              if (ct == NT - 1 && C::ZB_HALF && c >= 8) continue;
              *reinterpret_cast<u32x4*>(dst + C::zb_off(ct, 0, c, g)) = o[ct].hi;
              *reinterpret_cast<u32x4*>(dst + C::zb_off(ct, 1, c, g)) = o[ct].lo;
            }
          }
        }
      }
      WK_STAMP(2);
      __syncthreads();
      WK_STAMP(3);

      // ------------------------------------------------------------ the owned nodes, one at a time
#pragma unroll
      for (int s = 0; s < C::OWN; ++s) {
        if (s >= nown) continue;
        WK_RELANE();
        const int i = wave + WK_NW * s;
        const f32x4 pi4 = posT[i], p04 = posT[N + i];
        const float pi[3] = {pi4.x, pi4.y, pi4.z}, p0i[3] = {p04.x, p04.y, p04.z};
        // ===== primal of the node's edges: column c = partner jj
        const bool cval = c < NP;
        const int jj = cval ? c : 0, j = jj + (jj >= i ? 1 : 0);
        float dlt[3] = {0.f, 0.f, 0.f}, dl0[3] = {0.f, 0.f, 0.f}, rad = 0.f, e0 = 0.f;
        {
          const f32x4 pj4 = posT[j], q04 = posT[N + j];
          const float pj[3] = {pj4.x, pj4.y, pj4.z}, p0j[3] = {q04.x, q04.y, q04.z};
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            dlt[k] = pi[k] - pj[k];
            rad = fmaf(dlt[k], dlt[k], rad);
            dl0[k] = p0i[k] - p0j[k];
            e0 = fmaf(dl0[k], dl0[k], e0);
          }
        }
        float SQ;
        {
        float wr[8], we[8], g1[8], av[8], mp[8], pv[8], qv[8], ms[8];
        float al3[3], ep3[3], c3[3], phinv, pwr, pwe;
        wk_ld8(vecL + WV_WR * 32 + 8 * g, wr);
        wk_ld8(vecL + WV_WE * 32 + 8 * g, we);
        {
          float t0[8], t1[8], a1[8];
          wk_ld8(zaT + i * 32 + 8 * g, t0);
          wk_ld8(zbT + j * 32 + 8 * g, t1);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = t0[e] + t1[e] + fmaf(wr[e], rad, we[e] * e0);
            wk_silu_d(z, a1[e], g1[e]);
          }
          WkMat w2;
          wk_load(w2, mats, WM_W2, lane);
          WkFrag xa[1];
          wk_split8(a1, xa[0]);
          f32x4 z2[2][1] = {{zero4}, {zero4}};
          wk_gemm<1>(w2, xa, z2);
          float b2[8], watt[8], m[8], g2[8];
          wk_ld8(vecL + WV_B2 * 32 + 8 * g, b2);
          wk_ld8(vecL + WV_WATT * 32 + 8 * g, watt);
          const float z2v[8] = {z2[0][0].x, z2[0][0].y, z2[0][0].z, z2[0][0].w, z2[1][0].x, z2[1][0].y, z2[1][0].z, z2[1][0].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) wk_silu_d(fmaf(z2v[e], 1.f / WK_SW, b2[e]), m[e], g2[e]);
          float att = 1.0f;
          if (p.attention) att = fast_sigmoid(wk_gsum(wk_dot8(watt, m)) + vecL[WV_COUNT * 32]);
          const float datt = p.attention ? att * (1.0f - att) : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ms[e] = att * m[e];
            av[e] = att * g2[e];
            mp[e] = datt * m[e];
          }
          // coordinate head
          WkMat wc;
          wk_load(wc, mats, WM_WC1, lane);
          WkFrag xm[1];
          wk_split8(ms, xm[0]);
          f32x4 zc[2][1] = {{zero4}, {zero4}};
          wk_gemm<1>(wc, xm, zc);
          float bc1[8], wc2[8], ac[8], gcw[8];
          wk_ld8(vecL + WV_BC1 * 32 + 8 * g, bc1);
          wk_ld8(vecL + WV_WC2 * 32 + 8 * g, wc2);
          const float zcv[8] = {zc[0][0].x, zc[0][0].y, zc[0][0].z, zc[0][0].w, zc[1][0].x, zc[1][0].y, zc[1][0].z, zc[1][0].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gc;
            wk_silu_d(fmaf(zcv[e], 1.f / WK_SW, bc1[e]), ac[e], gc);
            gcw[e] = gc * wc2[e];
          }
          float cs = wk_gsum(wk_dot8(wc2, ac)), tau = 1.0f;
          if (p.tanh_on) {
            const float th = accurate_tanh(cs);
            tau = p.coord_scale * fmaf(-th, th, 1.0f);
            cs = th * p.coord_scale;
          }
          const float sq = sqrtf(rad + 1e-8f), inv = 1.0f / (sq + 1.0f), hsq = 0.5f / sq;
          float dhat[3] = {0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < DIM; ++k) dhat[k] = dlt[k] * inv;
          phinv = cs * inv;
          // adjoints: v_c = Wc1^T (gc o w_c2); p = g1 o W2^T (w_att o g2); q = g1 o W2^T (a o v_c) + p (m' . v_c)
          wk_load(wc, mats, WM_WC1T, lane);
          WkFrag xg[1];
          wk_split8(gcw, xg[0]);
          f32x4 zv[2][1] = {{zero4}, {zero4}};
          wk_gemm<1>(wc, xg, zv);
          float vc[8] = {zv[0][0].x, zv[0][0].y, zv[0][0].z, zv[0][0].w, zv[1][0].x, zv[1][0].y, zv[1][0].z, zv[1][0].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) vc[e] *= 1.f / WK_SW;
          wk_load(wc, mats, WM_W2T, lane);
          WkFrag xp[1], xq[1];
          {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = watt[e] * g2[e];
            wk_split8(t, xp[0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = av[e] * vc[e];
            wk_split8(t, xq[0]);
          }
          f32x4 zp[2][1] = {{zero4}, {zero4}}, zq[2][1] = {{zero4}, {zero4}};
          if (p.attention) wk_gemm<1>(wc, xp, zp);
          wk_gemm<1>(wc, xq, zq);
          const float zpv[8] = {zp[0][0].x, zp[0][0].y, zp[0][0].z, zp[0][0].w, zp[1][0].x, zp[1][0].y, zp[1][0].z, zp[1][0].w};
          const float zqv[8] = {zq[0][0].x, zq[0][0].y, zq[0][0].z, zq[0][0].w, zq[1][0].x, zq[1][0].y, zq[1][0].z, zq[1][0].w};
          const float mv = wk_gsum(wk_dot8(mp, vc));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pv[e] = g1[e] * zpv[e] * (1.f / WK_SW);
            qv[e] = fmaf(g1[e] * zqv[e], 1.f / WK_SW, pv[e] * mv);
          }
          pwr = wk_gsum(wk_dot8(pv, wr));
          pwe = wk_gsum(wk_dot8(pv, we));
          const float qwr = wk_gsum(wk_dot8(qv, wr)), qwe = wk_gsum(wk_dot8(qv, we));
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            c3[k] = dhat[k] * tau;
            al3[k] = fmaf(c3[k], qwr, -phinv * dhat[k] * hsq);
            ep3[k] = c3[k] * qwe;
          }
          // node sums over the valid columns -> the wave's scratch (read back by the node model / position update)
          {
            float agg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) agg[e] = wk_rowsum(cval ? ms[e] : 0.f);
            if (c == 0) wk_st8(nodeS + 8 * g, agg);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float t = wk_rowsum(cval ? dhat[k] * cs : 0.f);
              if (lane == 0) nodeS[32 + k] = t;
            }
          }
        }
        // per-node power of two for the coordinate rows: largest entry -> [2^10, 2^11)
        {
          float mx = fabsf(phinv);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            mx = fmaxf(mx, fabsf(al3[k]));
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(c3[k] * qv[e]));
          }
          mx = wk_wavemax(cval ? mx : 0.f);
          int ex = (int)((__float_as_uint(mx) >> 23) & 0xff);  // biased exponent (0 for zero / denormal; NaN -> 255)
          ex = 264 - ex;
          ex = ex < 1 ? 1 : (ex > 200 ? 200 : ex);
          SQ = __uint_as_float((unsigned)ex << 23);
        }
        // factor table of the edge loop
        if (cval) {
          float* f = FT + jj * C::FT_STRIDE;
          {  // (a, m') pairs in natural feature order: features 4g..4g+3 and 16+4g..
            const float v0[8] = {av[0], mp[0], av[1], mp[1], av[2], mp[2], av[3], mp[3]};
            const float v1[8] = {av[4], mp[4], av[5], mp[5], av[6], mp[6], av[7], mp[7]};
            wk_st8(f + C::FT_AM + 2 * (4 * g), v0);
            wk_st8(f + C::FT_AM + 2 * (16 + 4 * g), v1);
          }
          wk_st8(f + C::FT_B + 8 * g, g1);
          wk_st8(f + C::FT_P + 8 * g, pv);
          wk_st8(f + C::FT_Q + 8 * g, qv);
          if (g == 0) {  // per-edge scalars: coordinate rows' factors; what the S-product's coefficients need after the loop
            *reinterpret_cast<f32x4*>(f + C::FT_C3) = f32x4{SQ * c3[0], SQ * c3[1], SQ * c3[2], SQ * phinv};
            *reinterpret_cast<f32x4*>(f + C::FT_X) = f32x4{pwr, pwe, SQ * al3[0], SQ * al3[1]};
            *reinterpret_cast<f32x4*>(f + C::FT_X + 4) = f32x4{SQ * al3[2], SQ * ep3[0], SQ * ep3[1], SQ * ep3[2]};
          }
        }
        }
        wave_lds_fence();
        WK_STAMP(4);

        WK_RELANE();
        // ===== tangent: Acc = [d agg_i x (WK_SA WK_SZ) ; d pos_i update x (SQ WK_SZ)]
        f32x4 acc[3][NT];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[rb][ct] = zero4;
        float abar[3][8];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int e = 0; e < 8; ++e) abar[rb][e] = 0.f;
        if (!first) {
          float w2f[16];
          if (!last) {
            const f32x4* wp = reinterpret_cast<const f32x4*>(p.w2f + ((size_t)l * 64 + lane) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const f32x4 t = wp[q4];
              w2f[4 * q4] = t.x; w2f[4 * q4 + 1] = t.y; w2f[4 * q4 + 2] = t.z; w2f[4 * q4 + 3] = t.w;
            }
          }
          for (int e2 = 0; e2 < NP; ++e2) {
            const int j2 = e2 + (e2 >= i ? 1 : 0);
            const float* f = FT + e2 * C::FT_STRIDE;
            float qk[8];
            wk_ld8(f + C::FT_Q + 8 * g, qk);
            const float c3s = (c < 3) ? f[C::FT_C3 + (c & 3)] : 0.f;
            WkFrag A[3];
            {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                v[e] = c3s * qk[e];
                abar[2][e] += v[e];
              }
              wk_split8(v, A[2]);
            }
            if (!last) {
              float bk[8], pk[8];
              wk_ld8(f + C::FT_B + 8 * g, bk);
              wk_ld8(f + C::FT_P + 8 * g, pk);
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              const f32x2_t am0 = *reinterpret_cast<const f32x2_t*>(f + C::FT_AM + 2 * c);
              const f32x2_t am1 = *reinterpret_cast<const f32x2_t*>(f + C::FT_AM + 2 * (16 + c));
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) {
                const float a = rb ? am1.x : am0.x, mq = rb ? am1.y : am0.y;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  v[e] = fmaf(a, w2f[rb * 8 + e], (WK_SA * mq) * pk[e]) * bk[e];
                  abar[rb][e] += v[e];
                }
                wk_split8(v, A[rb]);
              }
            }
            const unsigned char* src = zb + j2 * C::ZB_NODE_C;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
              WkFrag Bf;
              Bf.hi = *reinterpret_cast<const u32x4*>(src + C::zb_off(ct, 0, c, g));
              Bf.lo = *reinterpret_cast<const u32x4*>(src + C::zb_off(ct, 1, c, g));
              if (!last) {
                acc[0][ct] = wk_mma3(A[0], Bf, acc[0][ct]);
                acc[1][ct] = wk_mma3(A[1], Bf, acc[1][ct]);
              }
              acc[2][ct] = wk_mma3(A[2], Bf, acc[2][ct]);
            }
          }
        }
        WK_STAMP(5);
        WK_RELANE();
        {
          // coefficients of dr / de: alpha = M w_r = a o (W2 (g1 o w_r)) + m' (p . w_r), eps likewise -- recomputed here from the
          // factor table (two small products per node) instead of being carried in 16 registers across the edge loop
          float al[8], ep[8], x8[8], phs;
          {
            const float* f = FT + jj * C::FT_STRIDE;
            float wr[8], we[8], g1[8], t[8];
            wk_ld8(vecL + WV_WR * 32 + 8 * g, wr);
            wk_ld8(vecL + WV_WE * 32 + 8 * g, we);
            wk_ld8(f + C::FT_B + 8 * g, g1);
            WkFrag xr[1], xe[1];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = g1[e] * wr[e];
            wk_split8(t, xr[0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = g1[e] * we[e];
            wk_split8(t, xe[0]);
            WkMat w2;
            wk_load(w2, mats, WM_W2, lane);
            f32x4 zr[2][1] = {{zero4}, {zero4}}, ze[2][1] = {{zero4}, {zero4}};
            wk_gemm<1>(w2, xr, zr);
            wk_gemm<1>(w2, xe, ze);
            float am0[8], am1[8];
            wk_ld8(f + C::FT_AM + 2 * (4 * g), am0);
            wk_ld8(f + C::FT_AM + 2 * (16 + 4 * g), am1);
            phs = f[C::FT_C3 + 3];
            wk_ld8(f + C::FT_X, x8);
            const float zrv[8] = {zr[0][0].x, zr[0][0].y, zr[0][0].z, zr[0][0].w, zr[1][0].x, zr[1][0].y, zr[1][0].z, zr[1][0].w};
            const float zev[8] = {ze[0][0].x, ze[0][0].y, ze[0][0].z, ze[0][0].w, ze[1][0].x, ze[1][0].y, ze[1][0].z, ze[1][0].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = e < 4 ? am0[2 * e] : am1[2 * (e - 4)], mq = e < 4 ? am0[2 * e + 1] : am1[2 * (e - 4) + 1];
              // invalid columns carry zero coefficients (they duplicate partner 0)
              al[e] = cval ? WK_SA * fmaf(a * zrv[e], 1.f / WK_SW, mq * x8[0]) : 0.f;
              ep[e] = cval ? WK_SA * fmaf(a * zev[e], 1.f / WK_SW, mq * x8[1]) : 0.f;
            }
            if (!cval) {
              phs = 0.f;
#pragma unroll
              for (int e = 2; e < 8; ++e) x8[e] = 0.f;
            }
          }
          wave_lds_fence();  // the factor table has been read: the R / E tables take its place
          WK_STAMP(6);
          WK_RELANE();
          // ---- R_i [35 rows][slot (j', k')]: Acc_i += R_i dPos_all  (dPos as B fragments every wave reads, K = 64)
          //   rows < 32: -2 Delta_ij',k' alpha_ij' ; column (i, k'): + sum_jj 2 Delta_ij,k' alpha_ij
          //   row 32 + k: -2 Delta_ij',k' alpha3_k - phi/(|d|+1) [k == k'] ; column (i, k'): + the sums
          {
            float* R = FT;
#pragma unroll
            for (int k2 = 0; k2 < DIM; ++k2) {
              const float m2 = -2.0f * dlt[k2];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = m2 * al[e];
                if (cval) R[wk_feat(8 * g + e) * C::R_STRIDE + 4 * j + k2] = v;
                const float sm = wk_rowsum(v);
                if (c == 0) R[wk_feat(8 * g + e) * C::R_STRIDE + 4 * i + k2] = -sm;
              }
#pragma unroll
              for (int k = 0; k < DIM; ++k) {  // coordinate rows (every lane reads the per-edge scalars of its column)
                const float w3 = fmaf(m2, x8[2 + k], (k == k2) ? -phs : 0.f);
                if (cval && g == 0) R[(32 + k) * C::R_STRIDE + 4 * j + k2] = w3;
                const float sm = wk_rowsum(w3);
                if (lane == 0) R[(32 + k) * C::R_STRIDE + 4 * i + k2] = -sm;
              }
            }
            if (lane < 35) {  // the pad slot of every node and the slots behind the last node
#pragma unroll
              for (int n2 = 0; n2 <= N; ++n2) R[lane * C::R_STRIDE + 4 * n2 + 3] = 0.f;
              R[lane * C::R_STRIDE + 4 * N] = 0.f; R[lane * C::R_STRIDE + 4 * N + 1] = 0.f; R[lane * C::R_STRIDE + 4 * N + 2] = 0.f;
            }
            wave_lds_fence();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              WkFrag As[3];
#pragma unroll
              for (int rb = 0; rb < 3; ++rb) {
                if (last && rb < 2) continue;
                const int row = rb < 2 ? 16 * rb + c : (c < 3 ? 32 + c : 0);
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (32 * ks + 8 * g < 4 * N + 4) wk_ld8(R + row * C::R_STRIDE + 32 * ks + 8 * g, v);
                if (rb == 2 && c >= 3) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                wk_split8(v, As[rb]);
              }
              const u32x4* src = reinterpret_cast<const u32x4*>(dposF + ks * C::ZB_NODE) + lane;
#pragma unroll
              for (int ct = 0; ct < NT; ++ct) {
                WkFrag Bs;
                Bs.hi = src[ct * 128];
                Bs.lo = src[ct * 128 + 64];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                  if (last && rb < 2) continue;
                  acc[rb][ct] = wk_mma3(As[rb], Bs, acc[rb][ct]);
                }
              }
            }
          }
          wave_lds_fence();
          WK_RELANE();
          // ---- E_i^T [direction (j', kk)][36 rows]: Acc_i += (WK_SZ) E_i   (d pos^0 = I: no product needed)
          {
            float* E = FT;
#pragma unroll
            for (int k2 = 0; k2 < DIM; ++k2) {
              const float m2 = (-2.0f * WK_SZ) * dl0[k2];
              float v[8], v3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = m2 * ep[e];
#pragma unroll
              for (int k = 0; k < DIM; ++k) v3[k] = m2 * x8[5 + k];
              if (cval) {
                float* dst = E + (j * DIM + k2) * C::E_STRIDE;
                *reinterpret_cast<f32x4*>(dst + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 16 + 4 * g) = f32x4{v[4], v[5], v[6], v[7]};
                if (g == 0) *reinterpret_cast<f32x4*>(dst + 32) = f32x4{v3[0], v3[1], v3[2], 0.f};
              }
              float sm[8], sm3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int e = 0; e < 8; ++e) sm[e] = -wk_rowsum(v[e]);
#pragma unroll
              for (int k = 0; k < DIM; ++k) sm3[k] = -wk_rowsum(v3[k]);
              if (c == 0) {
                float* dst = E + (i * DIM + k2) * C::E_STRIDE;
                *reinterpret_cast<f32x4*>(dst + 4 * g) = f32x4{sm[0], sm[1], sm[2], sm[3]};
                *reinterpret_cast<f32x4*>(dst + 16 + 4 * g) = f32x4{sm[4], sm[5], sm[6], sm[7]};
                if (g == 0) *reinterpret_cast<f32x4*>(dst + 32) = f32x4{sm3[0], sm3[1], sm3[2], 0.f};
              }
            }
            wave_lds_fence();
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
              const int d = 16 * ct + c;
              if (d < D) {
                const float* src = E + d * C::E_STRIDE;
                if (!last) {
                  acc[0][ct] += *reinterpret_cast<const f32x4*>(src + 4 * g);
                  acc[1][ct] += *reinterpret_cast<const f32x4*>(src + 16 + 4 * g);
                }
                if (g == 0) acc[2][ct] += *reinterpret_cast<const f32x4*>(src + 32);
              }
            }
          }
        }
        WK_STAMP(7);
        WK_RELANE();
        WkFrag xh[NT];
        if (!first) wk_frags<NT>(dHs[s], 1.0f, xh);
        if (!first) {  // (sum_j M_ij) (Wa dH_i)
          WkMat wa;
          wk_load(wa, mats, WM_WA, lane);
          f32x4 za[2][NT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) { za[0][ct] = zero4; za[1][ct] = zero4; }
          wk_gemm<NT>(wa, xh, za);
          WkFrag xa[NT];
          wk_frags<NT>(za, 1.0f, xa);
          WkFrag Ab[3];
#pragma unroll
          for (int rb = 0; rb < 3; ++rb) wk_split8(abar[rb], Ab[rb]);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
              if (last && rb < 2) continue;
              acc[rb][ct] = wk_mma3(Ab[rb], xa[ct], acc[rb][ct]);
            }
        }
        WK_STAMP(8);
        WK_RELANE();
        // position tangent leaving the layer (lanes g = 0 hold rows 32..35 of the third block)
        {
          const float un = 1.0f / (SQ * WK_SZ);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            if (g == 0) {
              const f32x4 o = dpn[i * (NT * 16) + 16 * ct + c];
              dpn[i * (NT * 16) + 16 * ct + c] = f32x4{fmaf(acc[2][ct].x, un, o.x), fmaf(acc[2][ct].y, un, o.y), fmaf(acc[2][ct].z, un, o.z), 0.f};
            }
          }
          if (lane == 0) posNew[i] = f32x4{pi[0] + nodeS[32], pi[1] + nodeS[33], DIM > 2 ? pi[2] + nodeS[34] : 0.f, 0.f};
        }
        if (!last) {  // node model (egnn_temp_conditioned.py:239-243, 284-291), primal (column s) and tangent
          float hs[8], ag[8];
          wk_ld8(nodeS + 8 * g, ag);
#pragma unroll
          for (int e = 0; e < 8; ++e) { hs[e] = (c == s) ? hq[e] : 0.f; ag[e] = (c == s) ? ag[e] : 0.f; }
          WkFrag xhp[1], xag[1];
          wk_split8(hs, xhp[0]);
          wk_split8(ag, xag[0]);
          WkMat wn;
          f32x4 zn[2][1] = {{zero4}, {zero4}};
          f32x4 dzn[2][NT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) { dzn[0][ct] = zero4; dzn[1][ct] = zero4; }
          wk_load(wn, mats, WM_WN1A, lane);
          wk_gemm<1>(wn, xhp, zn);
          if (!first) wk_gemm<NT>(wn, xh, dzn);
          wk_load(wn, mats, WM_WN1B, lane);
          wk_gemm<1>(wn, xag, zn);
          {
            f32x4 a2[2][NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) { a2[0][ct] = acc[0][ct]; a2[1][ct] = acc[1][ct]; }
            WkFrag xg[NT];
            wk_frags<NT>(a2, WK_ST / (WK_SA * WK_SZ), xg);
            wk_gemm<NT>(wn, xg, dzn);
          }
          float bn1[8], an[8], gn[8];
          wk_ld8(vecL + WV_BN1 * 32 + 8 * g, bn1);
          const float znv[8] = {zn[0][0].x, zn[0][0].y, zn[0][0].z, zn[0][0].w, zn[1][0].x, zn[1][0].y, zn[1][0].z, zn[1][0].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gq;
            wk_silu_d(fmaf(znv[e], 1.f / WK_SW, bn1[e]), an[e], gq);
            gn[e] = wk_rowsum((c == s) ? gq * (1.0f / WK_SW) : 0.f);  // column s's derivative to every column of the row
            if (c != s) an[e] = 0.f;
          }
          WkFrag xan[1];
          wk_split8(an, xan[0]);
          wk_load(wn, mats, WM_WN2, lane);
          f32x4 zo[2][1] = {{zero4}, {zero4}};
          wk_gemm<1>(wn, xan, zo);
          float bn2[8];
          wk_ld8(vecL + WV_BN2 * 32 + 8 * g, bn2);
          const float zov[8] = {zo[0][0].x, zo[0][0].y, zo[0][0].z, zo[0][0].w, zo[1][0].x, zo[1][0].y, zo[1][0].z, zo[1][0].w};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c == s) hq[e] += fmaf(zov[e], 1.f / WK_SW, bn2[e]);
          // tangent: dH_i += Wn2 (gn o (Wn1a dH_i + Wn1b d agg_i))
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            dzn[0][ct] *= f32x4{gn[0], gn[1], gn[2], gn[3]};
            dzn[1][ct] *= f32x4{gn[4], gn[5], gn[6], gn[7]};
          }
          WkFrag xz[NT];
          wk_frags<NT>(dzn, 1.0f, xz);
          f32x4 dho[2][NT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) { dho[0][ct] = zero4; dho[1][ct] = zero4; }
          wk_gemm<NT>(wn, xz, dho);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            dHs[s][0][ct] += dho[0][ct] * (1.0f / WK_SW);
            dHs[s][1][ct] += dho[1][ct] * (1.0f / WK_SW);
          }
        }
        wave_lds_fence();
        WK_STAMP(9);
      }
      __syncthreads();
      WK_STAMP(10);
      // ------------------------------------------------------------ publish the layer's results
      WK_RELANE();
#pragma unroll
      for (int s = 0; s < C::OWN; ++s) {
        if (s >= nown) continue;
        const int i = wave + WK_NW * s;
        if (lane == 0) posT[i] = posNew[i];
        publish_dpos(i);
        if (last && g == 0) {  // the diagonal entries of the trace: d pos^L_{i,k} / d x_{i,k} - 1
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            const int rel = 16 * ct + c - i * DIM;
            if (rel >= 0 && rel < DIM) {
              const f32x4 dp = dpn[i * (NT * 16) + 16 * ct + c];
              diag[i * DIM + rel] = (rel == 0 ? dp.x : (rel == 1 ? dp.y : dp.z)) - 1.0f;
            }
          }
        }
      }
    }
    WK_STAMP(11);
    __syncthreads();
    // ---------------------------------------------------------------- epilogue: trace and (optionally) the denoiser
    if (threadIdx.x == 0) {
      float sum = 0.f;
      for (int d = 0; d < D; ++d) sum += diag[d];
      float tr = fmaf(c_out * c_in, sum, (float)D * c_s);
      bool fin = __builtin_isfinite(tr);
      float mean[3] = {0.f, 0.f, 0.f};
      for (int n = 0; n < N; ++n) {
        const f32x4 a = posT[n], b = posT[N + n];
        mean[0] += a.x - b.x; mean[1] += a.y - b.y; mean[2] += a.z - b.z;
      }
      for (int k = 0; k < 3; ++k) { mean[k] /= (float)N; fin = fin && __builtin_isfinite(mean[k]); }
      for (int n = 0; n < N; ++n) {
        const f32x4 a = posT[n], b = posT[N + n];
        fin = fin && __builtin_isfinite(a.x - b.x) && __builtin_isfinite(a.y - b.y) && __builtin_isfinite(a.z - b.z);
      }
      if (p.mark) {
        p.mark[wid] = fin ? 0 : 1;
        if (!fin && p.bad_flag) *p.bad_flag = p.bad_seq;
      }
      if (fin) p.trace[wid] += tr;
      diag[48] = fin ? 1.0f : 0.f;
      diag[49] = mean[0]; diag[50] = mean[1]; diag[51] = mean[2];
    }
    __syncthreads();
    if (p.out && diag[48] != 0.f && threadIdx.x < D) {
      const int n = threadIdx.x / DIM, kk = threadIdx.x - n * DIM;
      const f32x4 a = posT[n], b = posT[N + n];
      const float F = (kk == 0 ? a.x - b.x : (kk == 1 ? a.y - b.y : a.z - b.z)) - diag[49 + kk];
      const long long gi = wid * D + threadIdx.x;
      p.out[gi] = fmaf(c_s, p.x[gi], c_out * F);
    }
  }
#ifdef PITA_WK_STAMPS
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 12; ++k) atomicAdd(&wk_dbg[k], wk_acc[k]);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// host side: packing (pita_egnn_create) and the launch (pita_egnn_jacobian_trace)
struct WkShape {
  int n, dim;
  void (*kernel)(WkParams);
  int lds_bytes;
};
#define PITA_WK_SHAPE(N, DIM) WkShape { N, DIM, egnn_div_walker_kernel<N, DIM>, WkCfg<N, DIM>::LDS_BYTES }
static const WkShape kWkShapes[] = {PITA_WK_SHAPE(13, 3)};
static const WkShape* find_wk_shape(int n, int dim) {
  for (const auto& s : kWkShapes)
    if (s.n == n && s.dim == dim) return &s;
  return nullptr;
}

// one allocation: [mats | w2f | vecs]
static size_t wk_mats_words(int L) { return (size_t)L * WM_COUNT * WK_MAT_W; }
static size_t wk_w2f_floats(int L) { return (size_t)L * 64 * 16; }
static size_t wk_vecs_floats(int L) { return WK_VEC_EMB_F + (size_t)L * WK_VEC_LAYER_F; }

int wk_pack_create(pita_egnn* net, const pita_egnn_config* cfg, const float* w) {
  net->d_wk = nullptr;
  if (!find_wk_shape(cfg->n_particles, cfg->n_dim) || cfg->hidden_nf != EH) return PITA_OK;
  const int H = EH, L = cfg->n_layers, nf = cfg->in_node_nf;
  const size_t nm = wk_mats_words(L), nw = wk_w2f_floats(L), nv = wk_vecs_floats(L);
  unsigned* hm = new unsigned[nm]();
  float* hw = new float[nw]();
  float* hv = new float[nv]();
  auto f16_bits = [](float v) { _Float16 hh = (_Float16)v; unsigned short u; memcpy(&u, &hh, 2); return (unsigned)u; };
  // A fragment of sc * M (32 x 32, row-major with leading dimension ld, columns col0..col0+31), k in slot order;
  // transposed: the fragment of sc * M^T
  auto pack = [&](unsigned* dst, const float* M, int ld, int col0, bool transposed) {
    for (int rb = 0; rb < 2; ++rb)
      for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 4; ++q) {
          unsigned pcs[2][2];
          for (int e2 = 0; e2 < 2; ++e2) {
            const int row = 16 * rb + (lane & 15), k = wk_feat(8 * (lane >> 4) + 2 * q + e2);
            const float v = WK_SW * (transposed ? M[k * ld + col0 + row] : M[row * ld + col0 + k]);
            const _Float16 v1 = (_Float16)v;
            pcs[e2][0] = f16_bits((float)v1);
            pcs[e2][1] = f16_bits(v - (float)v1);
          }
          for (int pc = 0; pc < 2; ++pc) dst[((rb * 2 + pc) * 64 + lane) * 4 + q] = pcs[0][pc] | (pcs[1][pc] << 16);
        }
  };
  auto pack_vec = [&](float* dst, const float* v, int stride) {
    for (int sl = 0; sl < 32; ++sl) dst[sl] = v[wk_feat(sl) * stride];
  };
  const float* q = w;
  const float* emb_w = q; q += H * nf;
  const float* emb_b = q; q += H;
  q += nf * H + nf;  // embedding_out: dead (egnn_temp_conditioned.py:80)
  for (int sl = 0; sl < 32; ++sl) {
    hv[sl] = emb_w[wk_feat(sl) * nf];
    hv[32 + sl] = nf == 2 ? emb_w[wk_feat(sl) * nf + 1] : 0.f;
    hv[64 + sl] = emb_b[wk_feat(sl)];
  }
  for (int l = 0; l < L; ++l) {
    const float* e0w = q; q += H * (2 * H + 2);
    const float* e0b = q; q += H;
    const float* e2w = q; q += H * H;
    const float* e2b = q; q += H;
    const float* n0w = q; q += H * 2 * H;
    const float* n0b = q; q += H;
    const float* n2w = q; q += H * H;
    const float* n2b = q; q += H;
    const float* c0w = q; q += H * H;
    const float* c0b = q; q += H;
    const float* c2w = q; q += H;
    const float* aw = nullptr; const float* ab = nullptr;
    if (cfg->attention) { aw = q; q += H; ab = q; q += 1; }
    unsigned* m = hm + (size_t)l * WM_COUNT * WK_MAT_W;
    pack(m + WM_WA * WK_MAT_W, e0w, 2 * H + 2, 0, false);
    pack(m + WM_WB * WK_MAT_W, e0w, 2 * H + 2, H, false);
    pack(m + WM_W2 * WK_MAT_W, e2w, H, 0, false);
    pack(m + WM_WC1 * WK_MAT_W, c0w, H, 0, false);
    pack(m + WM_WN1A * WK_MAT_W, n0w, 2 * H, 0, false);
    pack(m + WM_WN1B * WK_MAT_W, n0w, 2 * H, H, false);
    pack(m + WM_WN2 * WK_MAT_W, n2w, H, 0, false);
    pack(m + WM_WC1T * WK_MAT_W, c0w, H, 0, true);
    pack(m + WM_W2T * WK_MAT_W, e2w, H, 0, true);
    float* wf = hw + (size_t)l * 64 * 16;
    for (int lane = 0; lane < 64; ++lane)
      for (int rb = 0; rb < 2; ++rb)
        for (int e = 0; e < 8; ++e)
          wf[lane * 16 + rb * 8 + e] = WK_SA * e2w[(16 * rb + (lane & 15)) * H + wk_feat(8 * (lane >> 4) + e)];
    float* v = hv + WK_VEC_EMB_F + (size_t)l * WK_VEC_LAYER_F;
    pack_vec(v + WV_WR * 32, e0w + 2 * H, 2 * H + 2);
    pack_vec(v + WV_WE * 32, e0w + 2 * H + 1, 2 * H + 2);
    pack_vec(v + WV_B1 * 32, e0b, 1);
    pack_vec(v + WV_B2 * 32, e2b, 1);
    if (aw) pack_vec(v + WV_WATT * 32, aw, 1);
    pack_vec(v + WV_BC1 * 32, c0b, 1);
    pack_vec(v + WV_WC2 * 32, c2w, 1);
    pack_vec(v + WV_BN1 * 32, n0b, 1);
    pack_vec(v + WV_BN2 * 32, n2b, 1);
    v[WV_COUNT * 32] = ab ? ab[0] : 0.f;
  }
  const size_t bytes = nm * 4 + nw * 4 + nv * 4;
  hipError_t e = hipMalloc(&net->d_wk, bytes);
  if (e == hipSuccess) e = hipMemcpy(net->d_wk, hm, nm * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy((char*)net->d_wk + nm * 4, hw, nw * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy((char*)net->d_wk + nm * 4 + nw * 4, hv, nv * 4, hipMemcpyHostToDevice);
  delete[] hm;
  delete[] hw;
  delete[] hv;
  if (e != hipSuccess) {
    (void)hipFree(net->d_wk);
    net->d_wk = nullptr;
    return fail(PITA_EHIP, "pita_egnn_create: walker-resident trace pack: %s", hipGetErrorString(e));
  }
  return PITA_OK;
}

bool wk_available(const pita_egnn* net) {
  // opt-in while the kernel is being tuned (PITA_DIV_WALKER=1); read per call so that one process can A/B both paths
  const char* on = getenv("PITA_DIV_WALKER");
  return on && atoi(on) != 0 && net->d_wk && net->cfg.precision == 2 && net->cfg.n_layers >= 2 &&
         find_wk_shape(net->cfg.n_particles, net->cfg.n_dim) != nullptr;
}

// trace[b] += trace(J_x D) for the walkers the f16 path can represent; the others are marked (p.mark / bad_flag as the
// other fast kernels leave them) for the caller's bf16x3 repair launch
int wk_launch(pita_egnn* net, const float* h, const float* x, const float* beta, float* trace, float* out, long long B,
              int* mark, int* bad_flag, int bad_seq, hipStream_t st) {
  const WkShape* s = find_wk_shape(net->cfg.n_particles, net->cfg.n_dim);
  const int L = net->cfg.n_layers;
  WkParams p{};
  p.mats = reinterpret_cast<const unsigned*>(net->d_wk);
  p.w2f = reinterpret_cast<const float*>((const char*)net->d_wk + wk_mats_words(L) * 4);
  p.vecs = p.w2f + wk_w2f_floats(L);
  p.n_layers = L; p.in_nf = net->cfg.in_node_nf; p.attention = net->cfg.attention; p.tanh_on = net->cfg.tanh;
  p.feature_layout = net->cfg.feature_layout;
  p.coord_scale = net->cfg.coords_range / (float)L;
  p.B = B; p.h = h; p.x = x; p.beta = beta; p.trace = trace; p.out = out;
  p.mark = mark; p.bad_flag = bad_flag; p.bad_seq = bad_seq;
  PITA_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(s->kernel), (size_t)s->lds_bytes));
  long long grid = B < net->n_cu ? B : net->n_cu;  // one workgroup per CU (LDS), walkers strided over the grid
  if (const char* g = getenv("PITA_WK_GRID")) {  // development aid / test hook: the result must not depend on the grid
    const long long want = atoll(g);
    if (want > 0 && want <= 65535) grid = want;
  }
  hipLaunchKernelGGL(s->kernel, dim3((unsigned)grid), dim3(WK_NW * 64), s->lds_bytes, st, p);
  PITA_LAUNCH_CHECK();
  return PITA_OK;
}

}  // namespace pita


#ifdef PITA_WK_STAMPS
extern "C" int pita_wk_stamps(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(pita::wk_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pita::wk_dbg), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
