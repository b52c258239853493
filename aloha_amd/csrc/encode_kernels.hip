// encode_kernels.hip -- the CKKS encoder block (src/encoder) as FP64 HIP kernels for gfx950.
//
// Reference unit replaced: encoder_top (src/encoder/encoder_top.sv:12-52; controller.sv:225-235
// float pipeline, :589 modulus select, :636 signed -> residue), driven by the test-bench task
// run_encode (sim/top/top_noaxilite_tb.sv:419-448): N doubles = N/2 complex slots, (re, im)
// interleaved, become the plaintext's limbs in coefficient form; encode_post (the VP's NTT)
// follows.  The RTL uses a single-precision Xilinx FFT; this is the reference's software model
// instead (double precision, Delta = 2^38, rotation-group generator 3), whose convention is
// recovered from and checked against the tv/ encode pairs (oracle/ckks_encode.py has the
// derivation):
//     y_t  = z_j or conj(z_j)            (slot gather, host-built map, bit-reversed on load)
//     U    = FFT_n(y)                    (n = N/2 points, radix-2 DIT)
//     u_i  = U_i * zeta^-i * Delta / n   (twist)
//     m_i  = rint(Re u_i),  m_{i+n} = rint(Im u_i),  stored mod q_l for every limb l.
// One workgroup of 1024 threads per plaintext holds up to 8192 complex points (128 KiB) in LDS;
// N = 2^15 and 2^16 finish their last one or two stages in HBM.  This is floating point and off
// the integer hot path: parity is by tolerance (tests), not bit-exact.
#include "kernels.hpp"

namespace aloha {
namespace k {

namespace {

struct alignas(16) cplx { double x, y; };

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__device__ __forceinline__ u64 to_residue(double v, u64 q) {
  long long m = (long long)rint(v);                       // |v| < 2^62 checked on the host side (scale bound)
  long long r = m % (long long)q;
  return (u64)(r < 0 ? r + (long long)q : r);
}

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, unsigned bits) { return __brev(x) >> (32 - bits); }

// Stages len = 2 .. 2^lds_log of the n-point FFT on chunk `blockIdx.x` (2^lds_log points) of plaintext
// `blockIdx.y`, in LDS.  FINAL: the chunk is the whole transform -> twist, round, reduce, store
// limbs; otherwise the chunk goes to `work` for the remaining global stages.
template <bool FINAL>
__global__ __launch_bounds__(1024) void encode_lds_kernel(const double *__restrict__ re_im, EncodeTables tb,
                                                          cplx *__restrict__ work, u64 *__restrict__ out,
                                                          const u64 *__restrict__ moduli, uint32_t nlimbs,
                                                          unsigned log_slots, unsigned lds_log, double scale_over_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *s = reinterpret_cast<cplx *>(smem);
  const uint32_t n = 1u << log_slots, m = 1u << lds_log, tid = threadIdx.x;
  const uint32_t chunk = blockIdx.x, plain = blockIdx.y;
  const double *z = re_im + (size_t)plain * 2 * n;
  // bit-reversed gather: position p of the DIT input holds y[bitrev(p)]
  for (uint32_t i = tid; i < m; i += 1024) {
    const uint32_t p = chunk * m + i;
    const uint32_t e = tb.src[bitrev32(p, log_slots)];
    const uint32_t j = e & 0x7fffffffu;
    const double re = z[2 * j], im = z[2 * j + 1];
    s[i] = cplx{re, (e >> 31) ? -im : im};
  }
  __syncthreads();
  const cplx *w = reinterpret_cast<const cplx *>(tb.w);
  for (unsigned ls = 1; ls <= lds_log; ls++) {
    const uint32_t half = 1u << (ls - 1);
    for (uint32_t b = tid; b < m / 2; b += 1024) {
      const uint32_t pos = b & (half - 1), i0 = ((b >> (ls - 1)) << ls) + pos, i1 = i0 + half;
      const cplx t = cmul(s[i1], w[(size_t)pos << (log_slots - ls)]);
      const cplx a = s[i0];
      s[i0] = cplx{a.x + t.x, a.y + t.y};
      s[i1] = cplx{a.x - t.x, a.y - t.y};
    }
    __syncthreads();
  }
  if (!FINAL) {
    cplx *dst = work + (size_t)plain * n + (size_t)chunk * m;
    for (uint32_t i = tid; i < m; i += 1024) dst[i] = s[i];
    return;
  }
  const cplx *tw = reinterpret_cast<const cplx *>(tb.twist);
  const size_t N = (size_t)2 * n;
  for (uint32_t i = tid; i < n; i += 1024) {
    const cplx u = cmul(s[i], tw[i]);
    const double a = u.x * scale_over_n, b = u.y * scale_over_n;
    for (uint32_t l = 0; l < nlimbs; l++) {
      u64 *limb = out + ((size_t)plain * nlimbs + l) * N;
      limb[i] = to_residue(a, moduli[l]);
      limb[i + n] = to_residue(b, moduli[l]);
    }
  }
}

// one global radix-2 stage (len = 2^ls) for transforms larger than the LDS chunk
__global__ __launch_bounds__(256) void encode_stage_kernel(cplx *__restrict__ work, EncodeTables tb, unsigned log_slots,
                                                           unsigned ls) {
  const uint32_t n = 1u << log_slots, half = 1u << (ls - 1);
  cplx *s = work + (size_t)blockIdx.y * n;
  const cplx *w = reinterpret_cast<const cplx *>(tb.w);
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;       // butterfly index < n/2
  const uint32_t pos = b & (half - 1), i0 = ((b >> (ls - 1)) << ls) + pos, i1 = i0 + half;
  const cplx t = cmul(s[i1], w[(size_t)pos << (log_slots - ls)]);
  const cplx a = s[i0];
  s[i0] = cplx{a.x + t.x, a.y + t.y};
  s[i1] = cplx{a.x - t.x, a.y - t.y};
}

__global__ __launch_bounds__(256) void encode_finish_kernel(const cplx *__restrict__ work, EncodeTables tb,
                                                            u64 *__restrict__ out, const u64 *__restrict__ moduli,
                                                            uint32_t nlimbs, unsigned log_slots, double scale_over_n) {
  const uint32_t n = 1u << log_slots, i = blockIdx.x * 256 + threadIdx.x, plain = blockIdx.y;
  const cplx u = cmul(work[(size_t)plain * n + i], reinterpret_cast<const cplx *>(tb.twist)[i]);
  const double a = u.x * scale_over_n, b = u.y * scale_over_n;
  const size_t N = (size_t)2 * n;
  for (uint32_t l = 0; l < nlimbs; l++) {
    u64 *limb = out + ((size_t)plain * nlimbs + l) * N;
    limb[i] = to_residue(a, moduli[l]);
    limb[i + n] = to_residue(b, moduli[l]);
  }
}

// ---- decode: the inverse map, from one coefficient-form limb per plaintext ---------------------------
// u_i = c_i + i*c_{i+n} (centred lifts), x_i = u_i * zeta^i, y = IFFT_n(x) * n / Delta, slots from y by the same
// map.  IFFT(x) = conj(FFT(conj(x))) / n, so the stage kernel of the encoder is reused unchanged:
// work[p] = conj(x_{brv(p)}), log2 n radix-2 stages in HBM (off the hot path), then conj / Delta and the scatter.
__device__ __forceinline__ double centred_lift(u64 v, u64 q) {
  v = v >= q ? v - q : v;                                   // memory words are < 2q
  return v > (q >> 1) ? -(double)(q - v) : (double)v;
}

__global__ __launch_bounds__(256) void decode_prep_kernel(const u64 *__restrict__ limbs, size_t limb_stride, EncodeTables tb,
                                                          cplx *__restrict__ work, u64 q, unsigned log_slots) {
  const uint32_t n = 1u << log_slots, p = blockIdx.x * 256 + threadIdx.x, plain = blockIdx.y;
  const uint32_t i = bitrev32(p, log_slots);
  const u64 *limb = limbs + (size_t)plain * limb_stride;
  const cplx u{centred_lift(limb[i], q), centred_lift(limb[i + n], q)};
  const cplx tw = reinterpret_cast<const cplx *>(tb.twist)[i];            // zeta^-i; zeta^i is its conjugate
  const cplx x = cmul(u, cplx{tw.x, -tw.y});
  work[(size_t)plain * n + p] = cplx{x.x, -x.y};
}

__global__ __launch_bounds__(256) void decode_finish_kernel(const cplx *__restrict__ work, EncodeTables tb,
                                                            double *__restrict__ out, unsigned log_slots, double inv_scale) {
  const uint32_t n = 1u << log_slots, t = blockIdx.x * 256 + threadIdx.x, plain = blockIdx.y;
  const cplx r = work[(size_t)plain * n + t];
  const uint32_t e = tb.src[t], j = e & 0x7fffffffu;
  const double re = r.x * inv_scale, im = -r.y * inv_scale;               // y_t = conj(R_t) / Delta
  double *z = out + (size_t)plain * 2 * n;
  z[2 * j] = re;
  z[2 * j + 1] = (e >> 31) ? -im : im;
}

}  // namespace

hipError_t decode(hipStream_t st, const u64 *d_limbs, size_t limb_stride_words, const EncodeTables &tb, void *d_work,
                  double *d_out, u64 q, uint32_t nplain, unsigned logn, double scale, uint64_t *launches) {
  if (logn < 10 || logn > 16 || !nplain || !d_work) return hipErrorInvalidValue;
  const unsigned log_slots = logn - 1;
  const uint32_t n = 1u << log_slots;
  cplx *work = reinterpret_cast<cplx *>(d_work);
  hipLaunchKernelGGL(decode_prep_kernel, dim3(n / 256, nplain), dim3(256), 0, st, d_limbs, limb_stride_words, tb, work, q, log_slots);
  for (unsigned ls = 1; ls <= log_slots; ls++)
    hipLaunchKernelGGL(encode_stage_kernel, dim3(n / 2 / 256, nplain), dim3(256), 0, st, work, tb, log_slots, ls);
  hipLaunchKernelGGL(decode_finish_kernel, dim3(n / 256, nplain), dim3(256), 0, st, work, tb, d_out, log_slots, 1.0 / scale);
  if (launches) *launches += log_slots + 2;
  return hipGetLastError();
}

hipError_t encode(hipStream_t st, const double *d_re_im, const EncodeTables &tb, void *d_work, u64 *d_out,
                  const u64 *d_moduli, uint32_t nlimbs, uint32_t nplain, unsigned logn, double scale,
                  uint64_t *launches) {
  if (logn < 10 || logn > 16 || !nplain || !nlimbs) return hipErrorInvalidValue;
  const unsigned log_slots = logn - 1;
  const unsigned lds_log = log_slots < 13 ? log_slots : 13;          // <= 8192 points = 128 KiB of LDS
  const double scale_over_n = scale / (double)(1u << log_slots);
  const size_t lds_bytes = sizeof(cplx) << lds_log;
  cplx *work = reinterpret_cast<cplx *>(d_work);
  if (lds_log == log_slots) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_lds_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(encode_lds_kernel<true>, dim3(1, nplain), dim3(1024), lds_bytes, st, d_re_im, tb, work, d_out,
                       d_moduli, nlimbs, log_slots, lds_log, scale_over_n);
    if (launches) ++*launches;
    return hipGetLastError();
  }
  if (!work) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_lds_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encode_lds_kernel<false>, dim3(1u << (log_slots - lds_log), nplain), dim3(1024), lds_bytes, st,
                     d_re_im, tb, work, d_out, d_moduli, nlimbs, log_slots, lds_log, scale_over_n);
  if (launches) ++*launches;
  for (unsigned ls = lds_log + 1; ls <= log_slots; ls++) {
    hipLaunchKernelGGL(encode_stage_kernel, dim3((1u << log_slots) / 2 / 256, nplain), dim3(256), 0, st, work, tb,
                       log_slots, ls);
    if (launches) ++*launches;
  }
  hipLaunchKernelGGL(encode_finish_kernel, dim3((1u << log_slots) / 256, nplain), dim3(256), 0, st, work, tb, d_out,
                     d_moduli, nlimbs, log_slots, scale_over_n);
  if (launches) ++*launches;
  return hipGetLastError();
}

}  // namespace k
}  // namespace aloha
