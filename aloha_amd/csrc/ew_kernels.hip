// ew_kernels.hip -- element-wise modular ALU, automorphism and roll kernels for gfx950.
//
// Reference units replaced: modalu add/sub/mul/mod paths (src/vp/vxu/modalu.sv:196-249),
// the vaut/vroli address math + 14-stage routing network (src/vp/vxu/vxu_lane.sv:594-598,
// src/vp/iconn/iconn_top.sv:72-133).  All are HBM-bound streams: 16 B per lane per access
// (two coefficients), grid-strided, nothing staged through LDS.  The permutations are written
// as gathers so the stores stay coalesced.
#include "kernels.hpp"
#include "modarith.hpp"

namespace aloha {
namespace k {

namespace {

constexpr uint32_t kMaxGridY = 65535;       // batched kernels carry the job index in gridDim.y

struct alignas(16) u64x2 { u64 x, y; };

template <int OP>
__device__ __forceinline__ u64 ew_one(u64 a, u64 b, u64 s, u64 sp, const ModCtx &m) {
  const u64 q = m.q;
  // every ALU input passes ONE conditional subtraction (modalu.sv:44-46)
  a = csub(a, q);
  if (OP == EW_MUL_VV || OP == EW_ADD_VV || OP == EW_SUB_VV) b = csub(b, q);
  switch (OP) {
    case EW_MUL_VV: return mulmod_barrett(a, b, q, m.mu);
    case EW_MUL_VS: return csub(mul_shoup_2q(a, s, sp, q), q);
    case EW_ADD_VV: return addmod(a, b, q);
    case EW_ADD_VS: return addmod(a, s, q);
    case EW_SUB_VV: return submod(a, b, q);
    case EW_SUB_VS: return submod(a, s, q);
    case EW_SUB_SV: return submod(s, a, q);
    case EW_MOD:    return csub(a, q);      // a*1 through the Barrett unit (modalu.sv:164)
    case EW_CP:     return csub(a, q);      // a + 0 then the adder's conditional subtract (expander.v:396-417)
  }
  return 0;
}

template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(u64x2 *__restrict__ d, const u64x2 *__restrict__ a,
                                                 const u64x2 *__restrict__ b, u64 s, u64 sp, ModCtx m,
                                                 uint32_t n2) {
  constexpr bool VV = (OP == EW_MUL_VV || OP == EW_ADD_VV || OP == EW_SUB_VV);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    u64x2 va = a[i];
    u64x2 vb = VV ? b[i] : u64x2{0, 0};
    u64x2 r;
    r.x = ew_one<OP>(va.x, vb.x, s, sp, m);
    r.y = ew_one<OP>(va.y, vb.y, s, sp, m);
    d[i] = r;
  }
}

__global__ __launch_bounds__(256) void vaut_kernel(u64 *__restrict__ out, const u64 *__restrict__ in,
                                                   uint32_t n, u64 q, uint32_t kk, uint32_t kinv) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    uint32_t i = (j * kinv) & (n - 1);                 // the source index that lands on j
    uint32_t t = (i * kk) & (2 * n - 1);               // i*k mod 2N  (t mod N == j)
    u64 v = in[i];
    out[j] = (t >= n) ? q - v : v;                     // raw q - x: a flipped 0 is stored as q (vxu_lane.sv:598)
  }
}

__global__ __launch_bounds__(256) void vautn_kernel(u64 *__restrict__ out, const u64 *__restrict__ in, uint32_t n,
                                                    uint32_t kk) {
  const unsigned sh = __clz(n) + 1;
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const uint32_t e = ((2 * (__brev(j) >> sh) + 1) * kk) & (2 * n - 1);
    out[j] = in[__brev((e - 1) >> 1) >> sh];
  }
}

// vautz (extension): what vaut stores for a sign-flipped zero is q (vxu_lane.sv:598); in place is fine (element-wise)
__global__ __launch_bounds__(256) void vautz_kernel(u64 *__restrict__ out, const u64 *__restrict__ in, uint32_t n, u64 q, uint32_t kk, uint32_t kinv) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const u64 v = in[j];
    const uint32_t i = (j * kinv) & (n - 1);
    out[j] = (v == 0 && ((i * kk) & (2 * n - 1)) >= n) ? q : v;
  }
}

__global__ __launch_bounds__(256) void vroli_kernel(u64 *__restrict__ out, const u64 *__restrict__ in,
                                                    uint32_t n, uint32_t s) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256)
    out[j] = in[(j + s) & (n - 1)];
}

// ---- batched forms: blockIdx.y selects the job, so every branch below is block-uniform ---------
__global__ __launch_bounds__(256) void ew_batch_kernel(const EwJob *__restrict__ jobs, uint32_t n2) {
  const EwJob jb = jobs[blockIdx.y];
  u64x2 *d = reinterpret_cast<u64x2 *>(jb.d);
  const u64x2 *a = reinterpret_cast<const u64x2 *>(jb.a);
  const u64x2 *b = reinterpret_cast<const u64x2 *>(jb.b);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    const u64x2 va = a[i];
    u64x2 vb{0, 0}, r;
#define ALOHA_EW_BODY(OPC, VV)                                              \
  case OPC:                                                                  \
    if (VV) vb = b[i];                                                       \
    r.x = ew_one<OPC>(va.x, vb.x, jb.s, jb.s_shoup, jb.m);                    \
    r.y = ew_one<OPC>(va.y, vb.y, jb.s, jb.s_shoup, jb.m);                    \
    break;
    switch (jb.op) {
      ALOHA_EW_BODY(EW_MUL_VV, true) ALOHA_EW_BODY(EW_MUL_VS, false) ALOHA_EW_BODY(EW_ADD_VV, true)
      ALOHA_EW_BODY(EW_ADD_VS, false) ALOHA_EW_BODY(EW_SUB_VV, true) ALOHA_EW_BODY(EW_SUB_VS, false)
      ALOHA_EW_BODY(EW_SUB_SV, false) ALOHA_EW_BODY(EW_MOD, false) ALOHA_EW_BODY(EW_CP, false)
      default: r = va; break;
    }
#undef ALOHA_EW_BODY
    d[i] = r;
  }
}

__global__ __launch_bounds__(256) void aut_batch_kernel(const AutJob *__restrict__ jobs, uint32_t n) {
  const AutJob jb = jobs[blockIdx.y];
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    if (jb.roll == 1) {
      jb.out[j] = jb.in[(j + jb.k) & (n - 1)];
    } else if (jb.roll == 3) {                             // vautz: flipped zeros become q
      const u64 v = jb.in[j];
      const uint32_t i = (j * jb.k_inv) & (n - 1);
      jb.out[j] = (v == 0 && ((i * jb.k) & (2 * n - 1)) >= n) ? jb.q : v;
    } else if (jb.roll == 2) {                             // vautn: out[j] = in[brv(((2 brv(j) + 1) k mod 2N - 1) / 2)]
      const unsigned sh = __clz(n) + 1;                    // 32 - log2 N
      const uint32_t e = ((2 * (__brev(j) >> sh) + 1) * jb.k) & (2 * n - 1);
      jb.out[j] = jb.in[__brev((e - 1) >> 1) >> sh];
    } else {
      const uint32_t i = (j * jb.k_inv) & (n - 1);
      const uint32_t t = (i * jb.k) & (2 * n - 1);
      const u64 v = jb.in[i];
      jb.out[j] = (t >= n) ? jb.q - v : v;
    }
  }
}

// ---- automorphism as two coalesced passes (N = 256*R, R >= 16) -----------------------------------
// Write i = 256 r + c and k = 256 k1 + k0 (mod N).  Then  i*k mod N = 256 r' + c'  with
//     c' = c*k0 mod 256                      (depends on the column only)
//     r' = r*k0 + h(c) mod R,   h(c) = c*k1 + floor(c*k0 / 256)
// so the permutation factors into a permutation of the columns inside every 2 KiB row (pass A,
// whole rows through LDS) followed by a per-column rotation-and-scaling of the rows (pass B, 16
// columns x R rows through LDS, in place).  Both passes move whole 128-byte segments; together
// they cost 32 N bytes of traffic for 16 N algorithmic, against the gather's 64 distinct cache
// lines per wave-instruction.  The sign (i*k mod 2N >= N -> q - x, vxu_lane.sv:598) is applied in
// pass B from the original index.
__global__ __launch_bounds__(256) void aut_rows_kernel(const AutJob *__restrict__ jobs, uint32_t n) {
  __shared__ u64 s[16][256];
  const AutJob jb = jobs[blockIdx.y];
  const uint32_t t = threadIdx.x, r0 = blockIdx.x * 16;
  const u64 *in = jb.in + (size_t)r0 * 256;
  u64 *out = jb.out + (size_t)r0 * 256;
  const uint32_t srcc = (t * jb.k_inv) & 255;                 // out[r][t] = in[r][t * k0^-1 mod 256]
#pragma unroll
  for (int j = 0; j < 16; j++) s[j][t] = __builtin_nontemporal_load(&in[j * 256 + t]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; j++) out[j * 256 + t] = s[j][srcc];
}

template <int RLOG>      // R = 2^RLOG rows (4..8)
__global__ __launch_bounds__(256) void aut_cols_kernel(const AutJob *__restrict__ jobs, uint32_t n) {
  constexpr uint32_t R = 1u << RLOG, PER = R / 16, PIT = R + 1;
  __shared__ u64 s[16 * PIT];
  const AutJob jb = jobs[blockIdx.y];
  const uint32_t col = threadIdx.x & 15, slow = threadIdx.x >> 4, cp = blockIdx.x * 16 + col;   // cp = c'
  u64 *p = jb.out + cp;
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) s[col * PIT + slow + 16 * j] = p[(size_t)(slow + 16 * j) << 8];
  __syncthreads();
  const uint32_t k0 = jb.k & 255, k1 = (jb.k >> 8) & (R - 1), kinv = jb.k_inv;
  const uint32_t c = (cp * kinv) & 255;                       // source column of c'
  const uint32_t h = (c * k1 + ((c * k0) >> 8)) & (R - 1);
  u64 v[PER];
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t rp = slow + 16 * j;                        // r'
    const uint32_t r = ((rp - h) * kinv) & (R - 1);           // r = (r' - h) * k0^-1 mod R
    const uint32_t i = (r << 8) + c;
    const bool neg = ((i * jb.k) & (2 * n - 1)) >= n;         // k is kept mod N, like the RTL's scalar (vxu_lane.sv:594)
    const u64 x = s[col * PIT + r];
    v[j] = neg ? jb.q - x : x;
  }
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) __builtin_nontemporal_store(v[j], &p[(size_t)(slow + 16 * j) << 8]);
}

__global__ __launch_bounds__(256) void dot_kernel(DotArgs g, uint32_t n2) {
  const u64 q = g.m.q, mu = g.m.mu;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    u64x2 acc{0, 0};
    if (g.init) {
      acc = reinterpret_cast<const u64x2 *>(g.init)[i];
      acc.x = csub(acc.x, q); acc.y = csub(acc.y, q);
    }
    for (int t = 0; t < g.nterms; t++) {
      if (!g.a[t]) { acc.x = addmod(acc.x, g.s[t], q); acc.y = addmod(acc.y, g.s[t], q); continue; }   // constant term
      const u64x2 va = reinterpret_cast<const u64x2 *>(g.a[t])[i];
      if (g.b[t]) {
        const u64x2 vb = reinterpret_cast<const u64x2 *>(g.b[t])[i];
        acc.x = addmod(acc.x, mulmod_barrett(csub(va.x, q), csub(vb.x, q), q, mu), q);
        acc.y = addmod(acc.y, mulmod_barrett(csub(va.y, q), csub(vb.y, q), q, mu), q);
      } else {                                               // scalar term: Shoup product (any 64-bit input, result < 2q)
        acc.x = addmod(acc.x, csub(mul_shoup_2q(va.x, g.s[t], g.sp[t], q), q), q);
        acc.y = addmod(acc.y, csub(mul_shoup_2q(va.y, g.s[t], g.sp[t], q), q), q);
      }
    }
    reinterpret_cast<u64x2 *>(g.d)[i] = acc;
  }
}

__global__ __launch_bounds__(256) void dot_batch_kernel(const DotArgs *__restrict__ jobs, uint32_t n2) {
  const DotArgs *g = &jobs[blockIdx.y];                      // wave-uniform: operand pointers come through scalar loads
  const u64 q = g->m.q, mu = g->m.mu;
  const int nterms = g->nterms;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    u64x2 acc{0, 0};
    if (g->init) {
      acc = reinterpret_cast<const u64x2 *>(g->init)[i];
      acc.x = csub(acc.x, q); acc.y = csub(acc.y, q);
    }
    auto term = [&](int t, const u64x2 &a) {                 // acc += a .* (vector operand | scalar of a vfqmul.vs term) | constant
      if (!g->a[t]) {
        acc.x = addmod(acc.x, g->s[t], q); acc.y = addmod(acc.y, g->s[t], q);
      } else if (g->b[t]) {                                         // (uniform branch: the job descriptor is wave-uniform)
        const u64x2 b = reinterpret_cast<const u64x2 *>(g->b[t])[i];
        acc.x = addmod(acc.x, mulmod_barrett(csub(a.x, q), csub(b.x, q), q, mu), q);
        acc.y = addmod(acc.y, mulmod_barrett(csub(a.y, q), csub(b.y, q), q, mu), q);
      } else {
        const u64 s = g->s[t], sp = g->sp[t];
        acc.x = addmod(acc.x, csub(mul_shoup_2q(a.x, s, sp, q), q), q);
        acc.y = addmod(acc.y, csub(mul_shoup_2q(a.y, s, sp, q), q), q);
      }
    };
    int t = 0;
    auto first = [&](int t) -> u64x2 { return g->a[t] ? reinterpret_cast<const u64x2 *>(g->a[t])[i] : u64x2{0, 0}; };
    for (; t + 1 < nterms; t += 2) {                         // two terms per trip: their loads are in flight together
      const u64x2 a0 = first(t), a1 = first(t + 1);
      term(t, a0);
      term(t + 1, a1);
    }
    if (t < nterms) term(t, first(t));
    reinterpret_cast<u64x2 *>(g->d)[i] = acc;
  }
}

// tensor product of two ciphertexts, one job per modulus (TensorJob): every input is read once
__global__ __launch_bounds__(256) void tensor_batch_kernel(const TensorJob *__restrict__ jobs, uint32_t n2) {
  const TensorJob *g = &jobs[blockIdx.y];                     // wave-uniform
  const u64 q = g->m.q, mu = g->m.mu;
  const u64x2 *A0 = reinterpret_cast<const u64x2 *>(g->a0), *A1 = reinterpret_cast<const u64x2 *>(g->a1);
  const u64x2 *B0 = reinterpret_cast<const u64x2 *>(g->b0), *B1 = reinterpret_cast<const u64x2 *>(g->b1);
  u64x2 *D0 = reinterpret_cast<u64x2 *>(g->d0), *D1 = reinterpret_cast<u64x2 *>(g->d1), *D2 = reinterpret_cast<u64x2 *>(g->d2);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    u64x2 a0 = A0[i], a1 = A1[i], b0 = B0[i], b1 = B1[i];
    a0.x = csub(a0.x, q); a0.y = csub(a0.y, q); a1.x = csub(a1.x, q); a1.y = csub(a1.y, q);   // every ALU input passes one conditional subtraction (modalu.sv:44-46)
    b0.x = csub(b0.x, q); b0.y = csub(b0.y, q); b1.x = csub(b1.x, q); b1.y = csub(b1.y, q);
    u64x2 r;
    r.x = mulmod_barrett(a0.x, b0.x, q, mu); r.y = mulmod_barrett(a0.y, b0.y, q, mu);
    D0[i] = r;
    r.x = addmod(mulmod_barrett(a0.x, b1.x, q, mu), mulmod_barrett(a1.x, b0.x, q, mu), q);
    r.y = addmod(mulmod_barrett(a0.y, b1.y, q, mu), mulmod_barrett(a1.y, b0.y, q, mu), q);
    D1[i] = r;
    r.x = mulmod_barrett(a1.x, b1.x, q, mu); r.y = mulmod_barrett(a1.y, b1.y, q, mu);
    D2[i] = r;
  }
}

// exact-quotient Shoup product with a wave-uniform constant, on the pinned multiplier forms: [0, 2q) for ANY a
__device__ __forceinline__ u64 shoup_2q_u(u64 a, u64 s, u64 sp, u64 nq) {
  return mullo2_acc<true>(a, s, shoup_qhat_exact<true>(a, sp), nq);
}

__global__ __launch_bounds__(256) void sum_group_kernel(const SumGroup *__restrict__ groups, const SumJob *__restrict__ jobs,
                                                        uint32_t n2) {
  const SumGroup *g = &groups[blockIdx.y];                   // wave-uniform: pointers and scalars come through scalar loads
  const int T = g->nterms;
  const uint32_t first = g->first, count = g->count;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    u64x2 v[kSumTerms];
#pragma unroll
    for (int t = 0; t < kSumTerms; t++)
      if (t < T) v[t] = reinterpret_cast<const u64x2 *>(g->a[t])[i];
    for (uint32_t j = 0; j < count; j++) {
      const SumJob *jb = &jobs[first + j];
      const u64 q = jb->m.q, nq = jb->m.nq;
      u64 ax = 0, ay = 0;
      if (T <= 4) {                                           // <= 4 products < 4q each (approximate quotient, 3 instructions fewer)
#pragma unroll
        for (int t = 0; t < 4; t++)
          if (t < T) {
            const u64 sc = jb->s[t], sp = jb->sp[t];
            ax += mul_shoup_4q<true>(v[t].x, sc, sp, nq);
            ay += mul_shoup_4q<true>(v[t].y, sc, sp, nq);
          }
      } else {                                                // <= 8 products < 2q each (exact quotient): < 16q < 2^64 either way
#pragma unroll
        for (int t = 0; t < kSumTerms; t++)
          if (t < T) {
            const u64 sc = jb->s[t], sp = jb->sp[t];
            ax += shoup_2q_u(v[t].x, sc, sp, nq);
            ay += shoup_2q_u(v[t].y, sc, sp, nq);
          }
      }
      const u32 kc = (u32)jb->m.kc;
      const u64 c = jb->cst;
      u64x2 r;
      r.x = addmod(canon_any(ax, q, nq, kc), c, q);
      r.y = addmod(canon_any(ay, q, nq, kc), c, q);
      u64x2 *dst = reinterpret_cast<u64x2 *>(jb->d) + i;
      __builtin_nontemporal_store(r.x, &dst->x);
      __builtin_nontemporal_store(r.y, &dst->y);
    }
  }
}

inline uint32_t grid_for(uint32_t work_items) {
  uint32_t g = (work_items + 255) / 256;
  return g > 2048 ? 2048 : (g ? g : 1);
}


// ---- measurement helpers: synthetic inputs and the VALU clock ------------------------------------
// Synthetic residues of one limb: the accepted candidates, in order, of the splitmix64 stream seeded with
// seed = seed_base + limb index -- output n (0-based) is mix(seed + (n + 1) * 0x9E3779B97F4A7C15), its top 60 bits
// are the candidate, accepted when < q (rejection sampling: uniform in [0, q)).  splitmix64 is counter-based, so the
// candidates are computed in parallel; a rejection shifts everything behind it, which a chunk detects with one
// __syncthreads_count and repairs with a scan (for the largest 60-bit primes a rejection happens once in ~10^10).
__device__ __forceinline__ u64 splitmix_out(u64 seed, u64 n) {
  u64 z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(1024) void fill_splitmix_kernel(u64 *__restrict__ base, const u64 *__restrict__ moduli,
                                                             uint32_t n, u64 seed_base, u64 first_index) {
  __shared__ uint32_t wave_rej[16];
  const uint32_t limb = blockIdx.x, t = threadIdx.x;
  u64 *out = base + (size_t)limb * n;
  const u64 q = moduli[limb], seed = seed_base + first_index + limb;
  uint32_t rej_before = 0;                     // rejected candidates in earlier chunks (uniform)
  for (u64 c0 = 0; c0 - rej_before < n && c0 < 64ull * n; c0 += 1024) {   // (the host admits acceptance rates >= 1/16; the cap makes the loop finite whatever it is given)
    const u64 cand = splitmix_out(seed, c0 + t) >> 4;
    const bool rej = cand >= q;
    const uint32_t nrej = __syncthreads_count(rej);
    uint32_t before = 0;                       // rejected candidates of this chunk in front of this thread
    if (nrej) {
      const u64 m = __builtin_amdgcn_ballot_w64(rej);
      if ((t & 63) == 0) wave_rej[t >> 6] = (uint32_t)__popcll(m);
      __syncthreads();
      for (uint32_t w = 0; w < (t >> 6); w++) before += wave_rej[w];
      before += (uint32_t)__popcll(m & ((1ull << (t & 63)) - 1));
      __syncthreads();
    }
    const u64 idx = c0 + t - rej_before - before;
    if (!rej && idx < n) out[idx] = cand;
    rej_before += nrej;
  }
}

// VALU clock under an integer-multiply load: every SIMD hosts 8 waves, each issuing `iters` x 32 independent
// v_mad_u64_u32 (8 chains of 4); at 8 waves per SIMD the pipeline is full, and one wave64 VALU instruction occupies a
// 16-lane SIMD for 4 cycles (tools/ubench_valu.hip), so cycles per SIMD = 8 * 32 * iters * 4.
__global__ __launch_bounds__(256) void valu_clock_kernel(u64 *sink, uint32_t iters) {
  u64 a[8];
  const uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = x + i;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = mad32(lo32(a[i]), y, a[i]);
  }
  u64 s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x123456789abcdefull) sink[0] = s;      // never true in practice: keeps the chains alive
}

}  // namespace

hipError_t ew(hipStream_t st, EwOp op, u64 *d, const u64 *a, const u64 *b, u64 s, u64 sp, ModCtx m,
              uint32_t n) {
  const uint32_t n2 = n / 2;
  dim3 g(grid_for(n2)), blk(256);
  auto D = reinterpret_cast<u64x2 *>(d);
  auto A = reinterpret_cast<const u64x2 *>(a);
  auto B = reinterpret_cast<const u64x2 *>(b);
  switch (op) {
#define ALOHA_EW_CASE(OPC) \
  case OPC: hipLaunchKernelGGL(ew_kernel<OPC>, g, blk, 0, st, D, A, B, s, sp, m, n2); break;
    ALOHA_EW_CASE(EW_MUL_VV) ALOHA_EW_CASE(EW_MUL_VS) ALOHA_EW_CASE(EW_ADD_VV) ALOHA_EW_CASE(EW_ADD_VS)
    ALOHA_EW_CASE(EW_SUB_VV) ALOHA_EW_CASE(EW_SUB_VS) ALOHA_EW_CASE(EW_SUB_SV) ALOHA_EW_CASE(EW_MOD)
    ALOHA_EW_CASE(EW_CP)
#undef ALOHA_EW_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t ew_batch(hipStream_t st, const EwJob *d_jobs, uint32_t njobs, uint32_t n) {
  if (!njobs) return hipSuccess;
  const uint32_t n2 = n / 2;
  uint32_t gx = grid_for(n2);
  if (njobs >= 64 && gx > 32) gx = 32;             // enough blocks already: keep per-job grids short
  for (uint32_t off = 0; off < njobs; off += kMaxGridY)
    hipLaunchKernelGGL(ew_batch_kernel, dim3(gx, std::min(kMaxGridY, njobs - off)), dim3(256), 0, st, d_jobs + off, n2);
  return hipGetLastError();
}

hipError_t aut_batch(hipStream_t st, const AutJob *d_jobs, uint32_t njobs, unsigned logn, bool two_pass) {
  if (!njobs) return hipSuccess;
  if (two_pass && njobs > kMaxGridY) two_pass = false;            // (a program holds < 1400 such jobs; the gather form below chunks)
  const uint32_t n = 1u << logn;
  if (two_pass && logn >= 12 && logn <= 16) {          // automorphisms only (no rolls), R = N/256 >= 16 rows
    hipLaunchKernelGGL(aut_rows_kernel, dim3(n >> 12, njobs), dim3(256), 0, st, d_jobs, n);
    switch (logn - 8) {
#define ALOHA_AUT_COLS(RL) case RL: hipLaunchKernelGGL(aut_cols_kernel<RL>, dim3(16, njobs), dim3(256), 0, st, d_jobs, n); break;
      ALOHA_AUT_COLS(4) ALOHA_AUT_COLS(5) ALOHA_AUT_COLS(6) ALOHA_AUT_COLS(7) ALOHA_AUT_COLS(8)
#undef ALOHA_AUT_COLS
    }
    return hipGetLastError();
  }
  uint32_t gx = grid_for(n);
  if (njobs >= 64 && gx > 64) gx = 64;
  for (uint32_t off = 0; off < njobs; off += kMaxGridY)           // (gridDim.y carries the job index)
    hipLaunchKernelGGL(aut_batch_kernel, dim3(gx, std::min(kMaxGridY, njobs - off)), dim3(256), 0, st, d_jobs + off, n);
  return hipGetLastError();
}

hipError_t dot(hipStream_t st, const DotArgs &args, uint32_t n) {
  const uint32_t n2 = n / 2;
  hipLaunchKernelGGL(dot_kernel, dim3(grid_for(n2)), dim3(256), 0, st, args, n2);
  return hipGetLastError();
}

hipError_t dot_batch(hipStream_t st, const DotArgs *d_jobs, uint32_t njobs, uint32_t n) {
  if (!njobs) return hipSuccess;
  const uint32_t n2 = n / 2;
  uint32_t gx = grid_for(n2);
  while (gx > 16 && (uint64_t)gx * njobs > 8192) gx /= 2;    // plenty of workgroups already: longer grid-stride loops instead
  for (uint32_t off = 0; off < njobs; off += kMaxGridY)
    hipLaunchKernelGGL(dot_batch_kernel, dim3(gx, std::min(kMaxGridY, njobs - off)), dim3(256), 0, st, d_jobs + off, n2);
  return hipGetLastError();
}

hipError_t tensor_batch(hipStream_t st, const TensorJob *d_jobs, uint32_t njobs, uint32_t n) {
  if (!njobs) return hipSuccess;
  const uint32_t n2 = n / 2;
  uint32_t gx = grid_for(n2);
  while (gx > 16 && (uint64_t)gx * njobs > 8192) gx /= 2;
  for (uint32_t off = 0; off < njobs; off += kMaxGridY)
    hipLaunchKernelGGL(tensor_batch_kernel, dim3(gx, std::min(kMaxGridY, njobs - off)), dim3(256), 0, st, d_jobs + off, n2);
  return hipGetLastError();
}

hipError_t sum_groups(hipStream_t st, const SumGroup *d_groups, uint32_t ngroups, const SumJob *d_jobs, uint32_t n) {
  if (!ngroups) return hipSuccess;
  const uint32_t n2 = n / 2;
  uint32_t gx = (n2 + 255) / 256;                            // one element pair per thread: the loop over outputs is the work
  if (gx > 2048) gx = 2048;
  for (uint32_t off = 0; off < ngroups; off += kMaxGridY)
    hipLaunchKernelGGL(sum_group_kernel, dim3(gx ? gx : 1, std::min(kMaxGridY, ngroups - off)), dim3(256), 0, st, d_groups + off, d_jobs, n2);
  return hipGetLastError();
}

hipError_t vaut(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 q, u64 kk, u64 kinv) {
  const uint32_t n = 1u << logn;
  hipLaunchKernelGGL(vaut_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, in, n, q, (uint32_t)kk,
                     (uint32_t)kinv);
  return hipGetLastError();
}

hipError_t vautz(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 q, u64 kk, u64 kinv) {
  const uint32_t n = 1u << logn;
  hipLaunchKernelGGL(vautz_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, in, n, q, (uint32_t)kk, (uint32_t)kinv);
  return hipGetLastError();
}

hipError_t vautn(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 kk) {
  const uint32_t n = 1u << logn;
  hipLaunchKernelGGL(vautn_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, in, n, (uint32_t)kk);
  return hipGetLastError();
}

hipError_t vroli(hipStream_t st, u64 *out, const u64 *in, unsigned logn, u64 s) {
  const uint32_t n = 1u << logn;
  hipLaunchKernelGGL(vroli_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, in, n, (uint32_t)s);
  return hipGetLastError();
}

hipError_t fill_splitmix(hipStream_t st, u64 *base, const u64 *d_moduli, uint32_t count, uint32_t n, u64 seed_base,
                         u64 first_index) {
  if (!count) return hipSuccess;
  hipLaunchKernelGGL(fill_splitmix_kernel, dim3(count), dim3(1024), 0, st, base, d_moduli, n, seed_base, first_index);
  return hipGetLastError();
}

hipError_t valu_clock(hipStream_t st, u64 *d_sink, uint32_t simds, uint32_t iters) {
  // 8 workgroups of 4 waves per CU (4 SIMDs): 8 waves per SIMD
  hipLaunchKernelGGL(valu_clock_kernel, dim3(simds / 4 * 8), dim3(256), 0, st, d_sink, iters);
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
