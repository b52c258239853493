// ntt_single.hip -- single-pass negacyclic NTT / iNTT for N = 2^12, 2^13, 2^14 (gfx950).
//
// A limb of these sizes (32 / 64 / 128 KiB) fits the CU's 160 KiB LDS, so the whole transform is
// ONE read and ONE write of the limb: HBM traffic = the algorithmic 16*N bytes, where the
// two-pass tiles of ntt_tile.hip move twice that.  N = 2^13 is the reference's own size (every
// tv/ vector, every shipped program), N = 2^12 / 2^14 are BASELINE configs 2 / 3.
// Same results as every other path: natural order in -> bit-reversed out, twiddle
// psi^bitrev(m+i) (tf_rom_generator.sv:28-63), canonical outputs.
//
// One workgroup of T = N/16 threads per limb, 16 values per thread in registers, the limb in LDS
// between steps.  The log2 N stages are cut into radix-16 register steps over descending groups
// of four index bits, plus E = log2 N - 12 trailing stages:
//     L0  registers = index bits [n-1 .. n-4]     e = r*T + t            loads (coalesced 8 B lanes)
//     L1  registers = bits [n-5 .. n-8]
//     L2  registers = bits [n-9 .. n-12]
//     L3  registers = top 4-E bits and low E bits e = rhi*(N >> (4-E)) + t*2^E + rlo
//                                                                         stores (8*2^E B per lane, coalesced)
// with an LDS exchange between consecutive layouts (three per transform).  Elements with equal top
// four index bits form a REGION of N/16 words; in L1 and L2 a thread's registers and the threads of a
// wave stay inside regions owned by that wave (16 / 32 / 64 consecutive threads per region), so the
// middle exchange is wave-private (no s_barrier, wave_lds_fence) and a wave only ever writes its own
// regions outside the first write phase: two workgroup barriers per transform.  LDS word of element
// e: e ^ ((e >> 4) & 31) [^ ((e >> 5) & 31) for n > 12] -- a bijection under which every
// wave-instruction of every layout touches 32 distinct 8-byte banks per half-wave (checked
// exhaustively; no padding, so the limb occupies exactly N*8 bytes: 5 / 2 / 1 workgroups per CU).
// Stage s uses twiddle entry 2^s + (e >> (n - s)) of the plain table: for the first step that
// depends on the register index only (scalar loads), afterwards on the thread's upper bits
// (vector loads, 16 B per lane).  The inverse runs the same steps backwards with the
// Gentleman-Sande butterflies and fuses N^-1 into the last one.
#include <atomic>
#include <cstdlib>

#include "kernels.hpp"
#include "modarith.hpp"
#include "ntt_device.hpp"

namespace aloha {
namespace k {

namespace {

template <int LOGN>
__device__ __forceinline__ uint32_t swz(uint32_t e) {
  return LOGN == 12 ? e ^ ((e >> 4) & 31) : e ^ ((e >> 4) & 31) ^ ((e >> 5) & 31);
}

// layouts: element index held in register r of thread t
template <int LOGN, int K>      // K = 0, 1, 2: registers = index bits [LOGN-1-4K .. LOGN-4-4K]
__device__ __forceinline__ uint32_t lay(uint32_t t, uint32_t r) {
  constexpr int P = LOGN - 4 - 4 * K;
  return ((t >> P) << (P + 4)) | (r << P) | (t & ((1u << P) - 1));
}
template <int LOGN>             // final layout
__device__ __forceinline__ uint32_t lay3(uint32_t t, uint32_t r) {
  constexpr int E = LOGN - 12;
  return (r >> E) * ((1u << LOGN) >> (4 - E)) + (t << E) + (r & ((1u << E) - 1));
}

// twiddles of a radix-16 step over stages 4K .. 4K+3 for a thread whose upper index bits are A:
// entry 2^(4K+ss) + A*2^ss + g of the plain (w, w') table
struct TwPlain {
  ctw_t tab;
  uint32_t base, a;              // base = 2^(4K)
  __device__ __forceinline__ tw_t operator()(int ss, int g) const { return load_tw(tab + (base << ss) + (a << ss) + g); }
};

template <int NPRE, bool HI>      // NPRE entries from registers: the first ones (forward) or the last ones (inverse: sub-stage 3 first)
struct TwPlainMix {
  const tw_t (&r)[15];
  TwPlain rest;
  __device__ __forceinline__ tw_t operator()(int ss, int g) const {
    const int idx = (1 << ss) - 1 + g;
    return (HI ? idx >= 15 - NPRE : idx < NPRE) ? r[idx] : rest(ss, g);
  }
};
template <int NPRE, bool HI>
__device__ __forceinline__ void preload_tw(tw_t (&r)[15], ctw_t tab, uint32_t base, uint32_t a) {
#pragma unroll
  for (int ss = 0; ss < 4; ss++)
#pragma unroll
    for (int g = 0; g < (1 << ss); g++) {
      const int idx = (1 << ss) - 1 + g;
      if (HI ? idx >= 15 - NPRE : idx < NPRE) {
        const gtw_t p = (gtw_t)tab + (base << ss) + (a << ss) + g;
        r[idx].w = p->w;
        r[idx].wp = p->wp;
      }
    }
}
// Per-lane twiddles of the NEXT radix-16 step issued before the current one, as many as the register budget of
// 4 waves per SIMD allows: forward 0 at N = 2^12 (121 VGPRs as it is), 7 above (N = 2^13: 20.3 -> 21.4 M NTT/s;
// 15 spills: 15.2 M); inverse: see ALOHA_SINGLE_PRE_INV.
#ifndef ALOHA_SINGLE_PRE
#define ALOHA_SINGLE_PRE (LOGN == 12 ? 0 : 7)
#endif
#ifndef ALOHA_SINGLE_PRE_INV
#define ALOHA_SINGLE_PRE_INV 4   // (0 / 4 / 8: N = 2^12 43.8 / 44.8 / 39.5, N = 2^13 19.5 / 19.9 / 19.4 M inverse NTT/s)
#endif

#ifndef ALOHA_SINGLE_WAVES
#define ALOHA_SINGLE_WAVES 4   // at least 4 waves per SIMD (<= 128 VGPRs): two workgroups per CU at N = 2^13
#endif
template <int LOGN, bool INV, bool SC = false>
__global__ __launch_bounds__((1 << LOGN) / 16) __attribute__((amdgpu_waves_per_eu(ALOHA_SINGLE_WAVES))) void ntt_single_kernel(const LimbJob *__restrict__ jobs,
                                                                       const PrimeDev *__restrict__ primes) {
  constexpr int N = 1 << LOGN, T = N / 16, E = LOGN - 12, NH = 16 >> E;     // NH = groups of 2^E registers in L3
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const uint32_t t = threadIdx.x;
  const LimbJob jb = jobs[blockIdx.x];
  const PrimeDev *pr = &primes[jb.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t tab = (ctw_t)(INV ? pr->tw.inv : pr->tw.fwd);
  cgptr_t src = (cgptr_t)jb.src;
  gptr_t dst = (gptr_t)jb.dst;
  const uint32_t a1 = t >> (LOGN - 8), a2 = t >> E;           // upper index bits of the thread in L1 / L2
  u64 v[16];

  auto exchange = [&](auto from, auto to, bool wave_private) {   // registers: layout `from` -> layout `to` (LDS word of register r)
#pragma unroll
    for (int r = 0; r < 16; r++) lds[from(r)] = v[r];
    if (wave_private) wave_lds_fence(); else __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[to(r)];
    if (wave_private) wave_lds_fence();                        // (the next write phase reuses the same words)
  };
  // LDS word of register r in each layout.  At N = 2^13 -- the reference's ring, every tv/ vector -- in CLOSED FORM: the swizzle
  // only touches index bits 0..4, so the word is (a per-lane base XOR a 5-bit constant of r) + (r's own bits, an instruction
  // immediate): 0 / 1 / 1 / 0 vector instructions per access for L0 / L1 / L2 / L3 where the compiler derived ~3 from
  // swz(lay(t, r)) -- 14 % of the kernel's instructions, in a kernel that is VALU-bound (round 4; the identities are checked
  // exhaustively against swz / lay in tests/test_lds_layout.py).
  const uint32_t cA0 = t ^ (t >> 4) ^ ((t >> 5) & 15);
  const uint32_t cB1 = (((t >> 5) << 9) | (t & 31)) ^ ((t >> 4) & 1) ^ (((t >> 5) & 1) << 4);
  const uint32_t cC2 = (((t >> 1) << 5) | (t & 1)) ^ (((t >> 1) & 15) << 1) ^ ((t >> 1) & 31);
  const uint32_t cA3 = (t << 1) ^ ((t >> 3) & 31) ^ ((t >> 4) & 31);
  auto L0 = [&](uint32_t r) -> uint32_t { if constexpr (LOGN == 13) return (cA0 ^ ((r & 1) << 4)) + (r << 9); else return swz<LOGN>(lay<LOGN, 0>(t, r)); };
  auto L1 = [&](uint32_t r) -> uint32_t { if constexpr (LOGN == 13) return (cB1 ^ (((r << 1) ^ r) & 31)) + (r << 5); else return swz<LOGN>(lay<LOGN, 1>(t, r)); };
  auto L2 = [&](uint32_t r) -> uint32_t { if constexpr (LOGN == 13) return cC2 ^ ((r << 1) ^ (r >> 3)); else return swz<LOGN>(lay<LOGN, 2>(t, r)); };
  auto L3 = [&](uint32_t r) -> uint32_t { if constexpr (LOGN == 13) return (cA3 ^ (r & 1)) + ((r >> 1) << 10); else return swz<LOGN>(lay3<LOGN>(t, r)); };

  if (!INV) {
    constexpr int B0 = FWD_IN, B1 = bound_after4(B0), B2 = bound_after4(B1), B3 = bound_after4(B2);
    constexpr int B4 = next_bound(B3), B5 = next_bound(B4);
    static_assert(B3 <= 16 && B4 <= 16 && B5 <= 16, "forward lazy range overflows");
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = ld_stream(src + r * T + t);
    tw_t tw1[15], tw2[15];
    preload_tw<ALOHA_SINGLE_PRE, false>(tw1, tab, 16, a1);
    fwd_radix16<true, B0>(v, TwPlain{tab, 1, 0}, q, nq);                       // stages 0..3
    exchange(L0, L1, false);
    preload_tw<ALOHA_SINGLE_PRE, false>(tw2, tab, 256, a2);
    fwd_radix16<false, B1>(v, TwPlainMix<ALOHA_SINGLE_PRE, false>{tw1, TwPlain{tab, 16, a1}}, q, nq);   // stages 4..7
    exchange(L1, L2, true);
    fwd_radix16<false, B2>(v, TwPlainMix<ALOHA_SINGLE_PRE, false>{tw2, TwPlain{tab, 256, a2}}, q, nq);  // stages 8..11
    exchange(L2, L3, false);
    const u64 fourq = q << 2;
    if (E == 1) {                                              // stage 12: pairs (2h, 2h+1), group e >> 1 = h*T + t
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const tw_t w = load_tw(tab + 4096 + h * T + t);
        if (need_corr(B3)) v[2 * h] = csub_u(v[2 * h], q * (u64)corr_t(B3));
        ct_bfly<false, false>(v[2 * h], v[2 * h + 1], w.w, w.wp, nq, fourq);
      }
    } else if (E == 2) {                                       // stage 12: bit 1 (group e >> 2 = h*T + t); stage 13: bit 0
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const tw_t w = load_tw(tab + 4096 + h * T + t);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (need_corr(B3)) v[4 * h + j] = csub_u(v[4 * h + j], q * (u64)corr_t(B3));
          ct_bfly<false, false>(v[4 * h + j], v[4 * h + j + 2], w.w, w.wp, nq, fourq);
        }
      }
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int j = 0; j < 2; j++) {                          // group e >> 1 = h*(N/8) + 2t + j
          const tw_t w = load_tw(tab + 8192 + h * (N / 8) + 2 * t + j);
          if (need_corr(B4)) v[4 * h + 2 * j] = csub_u(v[4 * h + 2 * j], q * (u64)corr_t(B4));
          ct_bfly<false, false>(v[4 * h + 2 * j], v[4 * h + 2 * j + 1], w.w, w.wp, nq, fourq);
        }
    }
    const u32 kc = (u32)pr->m.kc;
    if (E == 0) {
#pragma unroll
      for (int r = 0; r < 16; r++) st_stream(dst + lay3<LOGN>(t, r), canon_any(v[r], q, nq, kc));
    } else {                                                     // a lane's 2^E adjacent words leave as 16-byte stores
#pragma unroll
      for (int r = 0; r < 16; r += 2) st_stream2(dst + lay3<LOGN>(t, r), canon_any(v[r], q, nq, kc), canon_any(v[r + 1], q, nq, kc));
    }
  } else {
    const u64 fourq = q << 2;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = ld_stream(src + lay3<LOGN>(t, r));     // < 2^60 < 4q
    if (E == 2) {
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const tw_t w = load_tw(tab + 8192 + h * (N / 8) + 2 * t + j);
          gs_bfly<false>(v[4 * h + 2 * j], v[4 * h + 2 * j + 1], w.w, w.wp, nq, fourq);
        }
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const tw_t w = load_tw(tab + 4096 + h * T + t);
#pragma unroll
        for (int j = 0; j < 2; j++) gs_bfly<false>(v[4 * h + j], v[4 * h + j + 2], w.w, w.wp, nq, fourq);
      }
    } else if (E == 1) {
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const tw_t w = load_tw(tab + 4096 + h * T + t);
        gs_bfly<false>(v[2 * h], v[2 * h + 1], w.w, w.wp, nq, fourq);
      }
    }
    tw_t tw2[15], tw1[15];
    preload_tw<ALOHA_SINGLE_PRE_INV, true>(tw2, tab, 256, a2);
    exchange(L3, L2, false);
    preload_tw<ALOHA_SINGLE_PRE_INV, true>(tw1, tab, 16, a1);
    inv_radix16<false, E ? INV_MID : INV_IN, false>(v, TwPlainMix<ALOHA_SINGLE_PRE_INV, true>{tw2, TwPlain{tab, 256, a2}}, q, nq);   // stages 11..8
    exchange(L2, L1, true);
    inv_radix16<false, INV_MID, false>(v, TwPlainMix<ALOHA_SINGLE_PRE_INV, true>{tw1, TwPlain{tab, 16, a1}}, q, nq);                // stages 7..4
    exchange(L1, L0, false);
    const LastScale ls = last_scale<SC>(&jobs[blockIdx.x], pr);
    inv_radix16<true, INV_MID, true>(v, TwPlain{tab, 1, 0}, q, nq, ls.w, ls.ni, ls.nip);   // stages 3..0, times N^-1 (times the job's scalar)
#pragma unroll
    for (int r = 0; r < 16; r++) st_stream(dst + r * T + t, canon_last<SC>(v[r], q, ls.addc));
  }
}


// ---- N = 2^12 with EIGHT values per thread ---------------------------------------------------------------------------------------
// The kernel above keeps 16 values per thread: 121 VGPRs, four waves per SIMD.  Here a limb is 512 threads x 8 values and the
// twelve stages are four radix-8 register steps (ntt_device.hpp):
//     L0  registers = index bits 11..9    e = r * 512 + t               loads (coalesced 8 B lanes), wave-uniform twiddles
//     L1  registers = bits 8..6           e = w * 512 + r * 64 + l      (w = wave, l = lane) wave-uniform twiddles
//     L2  registers = bits 5..3           e = w * 512 + (l >> 3) * 64 + r * 8 + (l & 7)
//     L3  registers = bits 2..0           e = w * 512 + l * 8 + r       stores (64 B per lane)
// After L0 a wave owns the 512 elements with its three top bits: ONE workgroup barrier, the other two exchanges are wave-private
// (lay1..lay3 and the swizzle of ntt_device.hpp inside the wave's region); loads and stores are lane-contiguous (a last / first
// wave-private transpose: a lane's own 8 words stored directly are 64 partial lines per instruction, half the kernel's speed).
// 64-72 VGPRs: six to eight waves per SIMD, three or four workgroups per CU.  Jobs with a folded scalar (LimbJob::sc) keep the
// sixteen-value kernel.
template <bool INV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void ntt_single8_kernel(const LimbJob *__restrict__ jobs, const PrimeDev *__restrict__ primes) {
  constexpr int N = 4096, T = 512;
  __shared__ u64 lds[N];
  const uint32_t t = threadIdx.x, l = t & 63, w = uni(t >> 6);
  const LimbJob jb = jobs[blockIdx.x];
  const PrimeDev *pr = &primes[jb.prime];
  const u64 q = pr->m.q, nq = pr->m.nq;
  ctw_t tab = (ctw_t)(INV ? pr->tw.inv : pr->tw.fwd);
  cgptr_t src = (cgptr_t)jb.src;
  gptr_t dst = (gptr_t)jb.dst;
  u64 *my = lds + w * 512;
  u64 v[8];
  if (!INV) {
    constexpr int B0 = FWD_IN, B1 = bound_after3(B0), B2 = bound_after3(B1), B3 = bound_after3(B2), B4 = bound_after3(B3);
    static_assert(B1 <= 16 && B2 <= 16 && B3 <= 16 && B4 <= 16, "forward lazy range overflows");
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = ld_stream(src + r * T + t);
    tw_t tw2[7];
    preload7(tw2, tab, 64, w * 8 + (l >> 3));
    fwd_radix8<true, B0>(v, TwRow{tab, 1, 0}, q, nq);                                    // stages 0..2
#pragma unroll
    for (int r = 0; r < 8; r++) lds[r * 512 + swh(t)] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw1(l, r)];
    fwd_radix8<true, B1>(v, TwRow{tab, 8, w}, q, nq);                                    // stages 3..5
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw1(l, r)] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw2(l, r)];
    fwd_radix8<false, B2>(v, TwReg7{tw2}, q, nq);                                        // stages 6..8
    tw_t tw3[7];                                                 // (requested only now: with tw2 alive as well the kernel spills at 80 VGPRs)
    preload7(tw3, tab, 512, w * 64 + l);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw2(l, r)] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw3(l, r)];
    fwd_radix8<false, B3>(v, TwReg7{tw3}, q, nq);                                        // stages 9..11
    const u32 kc = (u32)pr->m.kc;
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = canon_any(v[r], q, nq, kc);
    // back to lane-contiguous words for the stores (a lane's own 8 words are 64 B apart from its neighbour's: as stores that is
    // 64 partial lines per instruction -- measured at HALF the kernel's speed)
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw3(l, r)] = v[r];
    wave_lds_fence();
    gptr_t o = dst + w * 512 + l;
#pragma unroll
    for (int r = 0; r < 8; r++) st_stream(o + r * 64, my[sw1(l, r)]);
  } else {
    tw_t tw3[7], tw2[7];
    preload7(tw3, tab, 512, w * 64 + l);
    cgptr_t i = src + w * 512 + l;
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw1(l, r)] = ld_stream(i + r * 64);               // < 2^60 < 4q; lane-contiguous loads
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw3(l, r)];
    wave_lds_fence();
    preload7(tw2, tab, 64, w * 8 + (l >> 3));
    inv_radix8<false>(v, TwReg7{tw3}, q, nq);                                              // stages 11..9
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw3(l, r)] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw2(l, r)];
    inv_radix8<false>(v, TwReg7{tw2}, q, nq);                                              // stages 8..6
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw2(l, r)] = v[r];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = my[sw1(l, r)];
    inv_radix8<true>(v, TwRow{tab, 8, w}, q, nq);                                          // stages 5..3
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) my[sw1(l, r)] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = lds[r * 512 + swh(t)];
    // stages 2, 1 as plain Gentleman-Sande butterflies, stage 0 with N^-1 folded into its two products
    const u64 fourq = q << 2;
    {
      const tw_t a = load_tw(tab + 4), b = load_tw(tab + 5), c = load_tw(tab + 6), d = load_tw(tab + 7);
      gs_bfly<true>(v[0], v[1], a.w, a.wp, nq, fourq); gs_bfly<true>(v[2], v[3], b.w, b.wp, nq, fourq);
      gs_bfly<true>(v[4], v[5], c.w, c.wp, nq, fourq); gs_bfly<true>(v[6], v[7], d.w, d.wp, nq, fourq);
      const tw_t e = load_tw(tab + 2), f = load_tw(tab + 3);
      gs_bfly<true>(v[0], v[2], e.w, e.wp, nq, fourq); gs_bfly<true>(v[1], v[3], e.w, e.wp, nq, fourq);
      gs_bfly<true>(v[4], v[6], f.w, f.wp, nq, fourq); gs_bfly<true>(v[5], v[7], f.w, f.wp, nq, fourq);
    }
    const LastScale ls = last_scale<false>(nullptr, pr);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 x = v[j], y = v[j + 4];
      v[j] = mul_shoup_4q<true>(x + y, ls.ni, ls.nip, nq);
      v[j + 4] = mul_shoup_4q<true>(x + fourq - y, ls.w.w, ls.w.wp, nq);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) st_stream(dst + r * T + t, canon_4q(v[r], q));
  }
}

template <int LOGN, bool INV, bool SC = false>
hipError_t launch_single(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count) {
  constexpr int N = 1 << LOGN;
  // once per instantiation and DEVICE (the attribute is per device, and handles of one process may live on different
  // devices): the call is not free and the launch path is hot.  A racing second thread at worst sets it twice.
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  const uint64_t bit = 1ull << (dev & 63);
  if (dev > 63 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    const void *fn = reinterpret_cast<const void *>(&ntt_single_kernel<LOGN, INV, SC>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, N * 8);
    if (e != hipSuccess) return e;
    if (dev <= 63) attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((ntt_single_kernel<LOGN, INV, SC>), dim3(count), dim3(N / 16), N * 8, st, d_jobs, d_primes);
  return hipGetLastError();
}

}  // namespace

hipError_t ntt_single_batch(hipStream_t st, const LimbJob *d_jobs, const PrimeDev *d_primes, uint32_t count, unsigned logn,
                            bool inverse, uint64_t *launches, bool scaled) {
  if (!count) return hipSuccess;
  hipError_t e;
  if (inverse && scaled) {
    switch (logn) {
      case 12: e = launch_single<12, true, true>(st, d_jobs, d_primes, count); break;
      case 13: e = launch_single<13, true, true>(st, d_jobs, d_primes, count); break;
      case 14: e = launch_single<14, true, true>(st, d_jobs, d_primes, count); break;
      default: return hipErrorInvalidValue;
    }
    if (launches) ++*launches;
    return e;
  }
  // N = 2^12: eight values per thread (43.5-43.9 M forward / 43.4-43.8 M inverse NTT/s against 41.3-41.7 / 42.0 for sixteen,
  // 8192 limbs of 16 primes, profiles/r03_small_n_radix8.txt); ALOHA_SINGLE16=1 selects the sixteen-value kernel (A/B timing)
  static const bool eight = getenv("ALOHA_SINGLE16") == nullptr;
  if (eight && logn == 12) {
    if (inverse) hipLaunchKernelGGL((ntt_single8_kernel<true>), dim3(count), dim3(512), 0, st, d_jobs, d_primes);
    else hipLaunchKernelGGL((ntt_single8_kernel<false>), dim3(count), dim3(512), 0, st, d_jobs, d_primes);
    if (launches) ++*launches;
    return hipGetLastError();
  }
  switch (logn * 2 + (inverse ? 1 : 0)) {
    case 24: e = launch_single<12, false>(st, d_jobs, d_primes, count); break;
    case 25: e = launch_single<12, true>(st, d_jobs, d_primes, count); break;
    case 26: e = launch_single<13, false>(st, d_jobs, d_primes, count); break;
    case 27: e = launch_single<13, true>(st, d_jobs, d_primes, count); break;
    case 28: e = launch_single<14, false>(st, d_jobs, d_primes, count); break;
    case 29: e = launch_single<14, true>(st, d_jobs, d_primes, count); break;
    default: return hipErrorInvalidValue;
  }
  if (launches) ++*launches;
  return e;
}

}  // namespace k
}  // namespace aloha
