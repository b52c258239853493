// vp_engine.cpp -- host side of libaloha_vp.so: the VP as an ISA machine on one MI355X.
//
// What of the reference lives here (and what does not):
//   * instruction ROM + decoder          src/vp/sequncer/inst_rom.v:32-35, expander.v:65-700
//   * sequential issue (one op in flight) src/vp/sequncer/seq_top.v:343-513  -> a HIP stream
//   * vector register file (32 x N u64)  src/vp/vxu/vxu_rfbank.sv:13        -> one HBM slab
//   * SPM / KSK memories                 src/mem_buf/spm.sv:12, ksk_mem.sv:12 -> two HBM slabs
//   * vle/vse address generation         src/vp/vmu/addr_gen.v:44, vp_top_full.sv:105-117
//   * twiddle ROM contents               sim/vp/tf_rom_generator/tf_rom_generator.sv:28-111
// The queues, FSMs, AXI DMA and pipeline timing of the RTL have no equivalent: semantics are
// plain sequential.  There is no CPU compute path: without a gfx950 device every
// entry point fails with ALOHA_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/aloha_vp.h"
#include "kernels.hpp"
#include "numtheory.hpp"

using aloha::k::LimbJob;
using aloha::k::ModCtx;
using aloha::k::PrimeDev;
using aloha::k::TileDesc;
using aloha::k::TwiddleDev;
typedef unsigned long long u64;

namespace {

constexpr uint32_t kLanes = 128;
constexpr uint32_t kMaxPrimes = aloha::k::kPrimeTableEntries;

struct PrimeSlot {
  uint64_t q = 0;
  unsigned logn = 0;
  PrimeDev dev{};
  u64 *d_tables = nullptr;       // one allocation: fwd | inv | fwd_blk | inv_blk
};

constexpr size_t kMaxPlans = 128;

struct PlanChunk {               // limbs [limb_begin, +limb_count) and their row-pass tiles
  uint32_t limb_begin = 0, limb_count = 0;
  uint32_t uni_off = 0, n_uni = 0, glb_off = 0, n_glb = 0;
};

struct TransformPlan {           // device-side description of one batched transform, cached
  std::vector<LimbJob> jobs;     // 2*count entries: {src,dst} then the in-place form {dst,dst}
  uint64_t key_hash = 0;         // hash of the first `count` jobs: the cache compares it before the job arrays themselves
  unsigned logn = 0;
  std::vector<PlanChunk> chunks; // empty => radix-2 stage launches only
  LimbJob *d_jobs = nullptr;
  TileDesc *d_tiles = nullptr;
#ifdef ALOHA_EXPERIMENTS   // single-launch forward transform (experiments/ntt_tile_knobs.hip): ticket list + control words
  uint64_t *d_tickets = nullptr;
  uint32_t *d_ctl = nullptr;
  uint32_t npos = 0;
#endif
  ~TransformPlan() {
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_tiles) (void)hipFree(d_tiles);
#ifdef ALOHA_EXPERIMENTS
    if (d_tickets) (void)hipFree(d_tickets);
    if (d_ctl) (void)hipFree(d_ctl);
#endif
  }
};

struct RowsKey {                 // arguments of aloha_vp_ntt_rows, for the argument-level cache
  int space = 0; uint64_t dst_row = 0, src_row = 0; uint32_t log_n = 0, count = 0;
  std::vector<uint64_t> moduli;
  std::vector<LimbJob> jobs;
};

}  // namespace

struct aloha_vp {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  uint32_t flags = 0;
  uint64_t spm_rows = 0, ksk_rows = 0;
  unsigned max_logn = 16;
  uint32_t iram_depth = ALOHA_IRAM_DEPTH;            // instruction ROM words (inst_rom.v:32-35: 4096; deeper on request)
  u64 *d_spm = nullptr, *d_ksk = nullptr, *d_vreg = nullptr, *d_scratch = nullptr;
  // architectural state (vxu_top.sv:106-118)
  uint64_t vl_bits = 0, q = 0, mu = 0;
  std::vector<uint32_t> inst;
  std::vector<uint64_t> imm;
  // primes / twiddles
  std::vector<PrimeSlot> slots;
  std::map<std::pair<uint64_t, unsigned>, uint32_t> slot_of;
  PrimeDev *d_primes = nullptr;
  // batched transforms
  std::vector<std::unique_ptr<TransformPlan>> plans;
  std::vector<RowsKey> rows_cache;
  std::vector<std::pair<std::vector<uint8_t>, void *>> blobs;   // cached device copies of batched job arrays
  uint32_t max_slots = kMaxPrimes;   // twiddle sets kept at once (ALOHA_MAX_TWIDDLE_SETS lowers it: tests of the generation reset)
  uint32_t chunk_limbs = 0;      // limbs per chunk of a batched transform (0 = whole batch; chunking did not pay)
#ifdef ALOHA_EXPERIMENTS   // variant builds only (python -m aloha_amd.build --variant): measured and rejected, DESIGN.md 3.2 / 5
  // single-launch forward transform at N = 2^16 (experiments/ntt_tile_knobs.hip): batches of at least fused_min limbs
  uint32_t fused_min = 0, fused_delay = 8, fused_wgs = 0;
  uint32_t *d_fused_status = nullptr;                // sticky: non-zero after a fused launch gave up on a bounded wait
  bool fused_stats = false;                          // ALOHA_NTT_FUSED_STATS: print the scheduler's wait counts after every fused launch
  bool fused_unchecked = false;                      // a fused launch ran since the status word was last read
  bool graphs_on = false;                            // ALOHA_RUN_GRAPH=1: replay recorded runs as hipGraphs (measured SLOWER on ROCm 7.2)
#endif
  // CKKS encoder: tables per ring size, staging buffers
  struct EncTab { aloha::k::EncodeTables t{}; void *alloc = nullptr; };
  std::map<unsigned, EncTab> enc_tables;
  void *d_enc_in = nullptr, *d_enc_work = nullptr;
  size_t enc_in_bytes = 0, enc_work_bytes = 0;
  // polynomial products at N = 2^14 as one launch (ntt_product.hip): scratch for the transforms' intermediates, the arrival
  // counters (two banks, used by alternate launches) with a sticky status word behind them; allocated at the first product
  u64 *d_prod_work = nullptr;
  uint32_t *d_prod_ctl = nullptr;
  uint8_t prod_tag[aloha::k::kProdMaxLimbs] = {0};   // the tag the words of each limb slot of d_prod_work carry (0: as allocated)
  bool prod_unchecked = false;                       // a product launch ran since the status word was last read
  bool prod_disabled = false;                        // sticky: a product launch of this handle once gave up on a wait -- every later product takes the five-launch form
  u64 *d_prod_fb = nullptr;                          // two limbs of scratch per product job for that form (allocated with d_prod_work)
  bool capturing = false;                            // variant builds: a recorded run is being captured into a hipGraph
  // dataflow execution of small-ring programs (run_flow below): value buffers beside the vector registers
  u64 *d_flow = nullptr;
  size_t flow_bufs = 0;                              // limbs of 2^max_logn words in d_flow
  bool flow_on = true;
  // Launch lists of whole program runs, replayed when the same program segment is run again with the same CSRs
  // and the same architectural state on entry: the sequencer is a pure function of those (no data-dependent
  // control flow in the ISA), so a repeated phase of a key-switch costs its kernel launches and nothing else.
  struct RunPlan {
    std::vector<uint32_t> inst;                      // the segment [pc, break]
    std::vector<uint64_t> imm;
    uint64_t csr[4] = {0, 0, 0, 0}, step = 0, vl_in = 0, q_in = 0, mu_in = 0, vl_out = 0, q_out = 0, mu_out = 0;
    uint32_t pc = 0;
    uint64_t epoch = 0, instructions = 0;
    std::vector<std::function<hipError_t()>> ops;
    uint64_t scratch_row = 0, scratch_rows = 0;      // the scratch window the run was sequenced under (aloha_vp_set_scratch)
    // Variant builds, ALOHA_RUN_GRAPH=1: from its second replay on a list becomes a hipGraph -- the launches captured once
    // from `ops` on the handle's stream, re-issued by ONE hipGraphLaunch.  Measured on ROCm 7.2 / MI355X it is SLOWER for
    // these 4-7 kernel graphs: ~50 us of host time per hipGraphLaunch against ~3 us per plain launch (rank 0's share of a
    // sharded key-switch 0.43 ms instead of 0.11 ms, profiles/r02_l_ks_probe_*.jsonl), so the launch list is what ships.
    hipGraphExec_t graph = nullptr;
    bool graph_failed = false;
    uint32_t replays = 0;
    uint64_t launches = 0;                           // kernel launches per replay (statistics of a graph launch)
    RunPlan() = default;
    RunPlan(const RunPlan &) = delete;
    RunPlan &operator=(const RunPlan &) = delete;
    RunPlan(RunPlan &&o) noexcept { *this = std::move(o); }
    RunPlan &operator=(RunPlan &&o) noexcept {
      if (this != &o) {
        if (graph) (void)hipGraphExecDestroy(graph);
        inst = std::move(o.inst); imm = std::move(o.imm); ops = std::move(o.ops);
        memcpy(csr, o.csr, sizeof csr);
        step = o.step; vl_in = o.vl_in; q_in = o.q_in; mu_in = o.mu_in; vl_out = o.vl_out; q_out = o.q_out; mu_out = o.mu_out;
        scratch_row = o.scratch_row; scratch_rows = o.scratch_rows;
        pc = o.pc; epoch = o.epoch; instructions = o.instructions;
        graph = o.graph; graph_failed = o.graph_failed; replays = o.replays; launches = o.launches;
        o.graph = nullptr;
      }
      return *this;
    }
    ~RunPlan() { if (graph) (void)hipGraphExecDestroy(graph); }
  };
  std::unordered_multimap<uint64_t, RunPlan> run_cache;
  RunPlan *rec = nullptr;                            // non-null while a run is being recorded
  uint64_t epoch = 0;                                // bumped whenever a cached device object a plan may point to is freed
  bool run_cache_on = true;
  uint64_t scratch_row = 0, scratch_rows = 0;        // aloha_vp_set_scratch: SPM rows whose contents are undefined after a run that writes them
  aloha_vp_stats stats{};
  std::string err;
};

namespace {

// Every entry point that touches HIP runs with the handle's device current and restores the caller's on return: in a
// process that sees several GPUs (torch switching devices, handles on different devices) allocations and launches would
// otherwise land on whatever device the calling thread had selected last.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const aloha_vp *vp);
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

aloha_status fail(aloha_vp *vp, aloha_status s, const char *fmt, ...) {
  if (vp) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    vp->err = buf;
  }
  return s;
}

DeviceGuard::DeviceGuard(const aloha_vp *vp) {
  if (!vp) return;
  if (hipGetDevice(&prev) == hipSuccess && prev != vp->device) switched = hipSetDevice(vp->device) == hipSuccess;
}

#define HIPCK(vp, expr)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(vp, ALOHA_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

// Every enqueue of a program run goes through here: executed now, and kept for replay while a run is recorded.
template <class F>
inline hipError_t issue(aloha_vp *vp, F f) {
  const hipError_t e = f();
  if (e == hipSuccess && vp->rec) vp->rec->ops.emplace_back(std::move(f));
  return e;
}

u64 *vreg(aloha_vp *vp, unsigned r) { return vp->d_vreg + ((size_t)r << vp->max_logn); }

aloha_status vl_logn(uint64_t vl_bits, unsigned max_logn, unsigned *logn) {
  if (vl_bits % 64) return ALOHA_ERR_VL;
  const uint64_t n = vl_bits / 64;
  for (unsigned l = 10; l <= 16; l++)
    if (n == (1ull << l)) {
      if (l > max_logn) return ALOHA_ERR_VL;
      *logn = l;
      return ALOHA_OK;
    }
  return ALOHA_ERR_VL;                               // ntt_fsm.sv:59-65 admits log2 N in 10..16
}

// primality is checked once per value and thread: programs set the same few moduli over and over
bool modulus_ok(uint64_t q) {
  if (!(q > (1ull << 59) && q < (1ull << 60))) return false;
  thread_local std::vector<uint64_t> known;
  for (uint64_t k : known) if (k == q) return true;
  if (!aloha::nt::is_prime(q)) return false;
  if (known.size() >= 4096) known.clear();
  known.push_back(q);
  return true;
}

ModCtx modctx(uint64_t q) { return ModCtx{q, 0 - q, aloha::nt::barrett_mu(q), (1ull << 59) / ((q >> 32) + 1)}; }

// The twiddle cache is bounded (max_slots <= kMaxPrimes sets of ~4 MiB at N = 2^16).  Slot indices live inside cached job
// arrays, plans and recorded runs, so sets are not evicted one by one: an entry point that may need `incoming` new sets
// and would overflow the cache starts a new GENERATION first -- the stream drains, every set is freed, every cache that
// could hold a slot index is dropped (the epoch invalidates recorded runs) -- and only then resolves its moduli.
aloha_status make_room_for_twiddles(aloha_vp *vp, size_t incoming) {
  if (vp->slots.size() + incoming <= vp->max_slots) return ALOHA_OK;
  if (incoming > vp->max_slots)
    return fail(vp, ALOHA_ERR_NOMEM, "one call needs %zu (q, N) twiddle sets, the cache holds %u", incoming, vp->max_slots);
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  for (auto &sl : vp->slots) if (sl.d_tables) (void)hipFree(sl.d_tables);
  vp->slots.clear();
  vp->slot_of.clear();
  vp->plans.clear();
  vp->rows_cache.clear();
  vp->run_cache.clear();
  vp->epoch++;
  vp->stats.twiddle_generations++;
  return ALOHA_OK;
}

// Twiddle set for (q, N): generated on first use, kept until the cache starts a new generation (above).
aloha_status get_slot(aloha_vp *vp, uint64_t q, unsigned logn, uint32_t *out) {
  auto key = std::make_pair(q, logn);
  auto it = vp->slot_of.find(key);
  if (it != vp->slot_of.end()) { *out = it->second; return ALOHA_OK; }
  if (vp->slots.size() >= vp->max_slots) return fail(vp, ALOHA_ERR_NOMEM, "more than %u (q, N) twiddle sets in one call", vp->max_slots);
  if ((q - 1) % (2ull << logn)) return fail(vp, ALOHA_ERR_ROOT, "q = %#llx is not 1 mod 2N (N = 2^%u)", (u64)q, logn);
  aloha::nt::TwiddleTables tt;
  if (!aloha::nt::make_twiddles(q, logn, &tt)) return fail(vp, ALOHA_ERR_ROOT, "no primitive 2N-th root mod %#llx", (u64)q);
  const uint32_t n = 1u << logn, n1 = logn - 8, rows = 1u << n1;
  const size_t blk_words = (size_t)(1 + rows) * 256 * 2;
  // blocked layout for the tile kernels: block 0 = W[0..255]; block 1+r, entry 2^s'+i =
  // W[2^(n1+s') + r*2^s' + i]  (ntt_tile.hip header)
  std::vector<uint64_t> host(2 * (size_t)n * 2 + 2 * blk_words, 0);
  uint64_t *fwd = host.data(), *inv = fwd + 2 * (size_t)n, *fblk = inv + 2 * (size_t)n, *iblk = fblk + blk_words;
  memcpy(fwd, tt.fwd.data(), 2 * (size_t)n * 8);
  memcpy(inv, tt.inv.data(), 2 * (size_t)n * 8);
  for (int dir = 0; dir < 2; dir++) {
    const uint64_t *tab = dir ? inv : fwd;
    uint64_t *blk = dir ? iblk : fblk;
    for (uint32_t L = 0; L < 256 && L < n; L++) { blk[2 * L] = tab[2 * L]; blk[2 * L + 1] = tab[2 * L + 1]; }
    for (uint32_t r = 0; r < rows; r++)
      for (unsigned s = 0; s < 8; s++)
        for (uint32_t i = 0; i < (1u << s); i++) {
          const size_t srci = ((size_t)1 << (n1 + s)) + ((size_t)r << s) + i;
          const size_t dsti = (size_t)(1 + r) * 256 + (1u << s) + i;
          blk[2 * dsti] = tab[2 * srci];
          blk[2 * dsti + 1] = tab[2 * srci + 1];
        }
  }
  PrimeSlot ps;
  ps.q = q; ps.logn = logn;
  HIPCK(vp, hipMalloc(&ps.d_tables, host.size() * 8));
  struct Undo { u64 *&p; bool armed = true; ~Undo() { if (armed && p) { (void)hipFree(p); p = nullptr; } } } undo{ps.d_tables};   // no leak on the error paths below
  HIPCK(vp, hipMemcpyAsync(ps.d_tables, host.data(), host.size() * 8, hipMemcpyHostToDevice, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));      // `host` goes out of scope
  ps.dev.m = modctx(q);
  ps.dev.tw.fwd = ps.d_tables;
  ps.dev.tw.inv = ps.d_tables + 2 * (size_t)n;
  ps.dev.tw.fwd_blk = ps.d_tables + 4 * (size_t)n;
  ps.dev.tw.inv_blk = ps.d_tables + 4 * (size_t)n + blk_words;
  ps.dev.tw.n_inv = tt.n_inv;
  ps.dev.tw.n_inv_shoup = tt.n_inv_shoup;
  ps.dev.tw.w1n_inv = tt.w1n_inv;
  ps.dev.tw.w1n_inv_shoup = tt.w1n_inv_shoup;
  const uint32_t idx = (uint32_t)vp->slots.size();
  HIPCK(vp, hipMemcpyAsync(vp->d_primes + idx, &ps.dev, sizeof(PrimeDev), hipMemcpyHostToDevice, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  undo.armed = false;
  vp->slots.push_back(ps);
  vp->slot_of[key] = idx;
  vp->stats.twiddle_builds++;
  *out = idx;
  return ALOHA_OK;
}

// ---- admission of one-launch polynomial products (ntt_product.hip) --------------------------------------------------------------
// A product launch's later phases wait INSIDE the grid for its earlier ones.  Progress needs every XCD to dispatch each queue's
// workgroups in order (observed, not promised by HIP); launches of several handles on different streams keep that property per
// queue, but they share the chip's workgroup slots with one another's WAITING workgroups, and a wait that outlasts its bound
// spoils an output.  So at most ONE handle per device has product launches in flight: a handle that finds another handle's launch
// unfinished runs the same product as five ordinary launches (transforms, product, inverse transforms: the form of rounds 1-2,
// word-identical results).  So does, for good, a handle whose status word was ever set (prod_disabled).
struct ProdGate {
  std::mutex mu;
  const aloha_vp *owner = nullptr;
  hipEvent_t done = nullptr;                         // recorded behind the owner's latest product launch
  bool pending = false;
};
ProdGate &prod_gate(int device) {
  static ProdGate gates[64];
  return gates[(unsigned)device % 64];
}
// may `vp` launch a one-launch product now?  (called with the handle's device current)
bool prod_admit(const aloha_vp *vp) {
  ProdGate &g = prod_gate(vp->device);
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.owner && g.owner != vp && g.pending) {
    if (hipEventQuery(g.done) != hipSuccess) { (void)hipGetLastError(); return false; }   // another handle's product is still in flight
    g.pending = false;
  }
  g.owner = vp;
  return true;
}
void prod_launched(const aloha_vp *vp, hipStream_t st) {
  ProdGate &g = prod_gate(vp->device);
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.done && hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) { g.done = nullptr; (void)hipGetLastError(); return; }
  if (g.owner == vp && hipEventRecord(g.done, st) == hipSuccess) g.pending = true;
}
void prod_release(const aloha_vp *vp) {              // the handle goes away (its stream has drained)
  ProdGate &g = prod_gate(vp->device);
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.owner == vp) { g.owner = nullptr; g.pending = false; }
}

// After a synchronisation: did a fused transform give up on one of its bounded waits?  (It cannot in a correct
// schedule -- see ntt_fused_fwd_kernel -- but a wait that can be bounded is, and the bound is reported.)
aloha_status check_product(aloha_vp *vp) {
  if (!vp->prod_unchecked) return ALOHA_OK;
  uint32_t st = 0;
  HIPCK(vp, hipMemcpyAsync(&st, vp->d_prod_ctl, 4, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  vp->prod_unchecked = false;
  if (!st) return ALOHA_OK;
  HIPCK(vp, hipMemsetAsync(vp->d_prod_ctl, 0, 64, vp->stream));
  vp->prod_disabled = true;                          // sticky: this handle's later products run as five ordinary launches
  return fail(vp, ALOHA_ERR_DEVICE, "a one-launch polynomial product gave up on a bounded wait (%u waits): its output is incomplete; "
                                    "the operands are intact (in-place products never take that form) and later products of this handle run as five launches", st);
}

aloha_status check_fused(aloha_vp *vp) {
#ifndef ALOHA_EXPERIMENTS
  return check_product(vp);
#else
  { aloha_status sp = check_product(vp); if (sp) return sp; }
  if (!vp->fused_unchecked) return ALOHA_OK;
  uint32_t st = 0;
  HIPCK(vp, hipMemcpyAsync(&st, vp->d_fused_status, 4, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  vp->fused_unchecked = false;
  if (!st) return ALOHA_OK;
  HIPCK(vp, hipMemsetAsync(vp->d_fused_status, 0, 4, vp->stream));
  return fail(vp, ALOHA_ERR_DEVICE, "fused N=2^16 transform gave up on a bounded wait (%u waits): its output is incomplete", st);
#endif
}

// Batched transform of `count` limbs described by (src, dst, slot); the workhorse behind both
// the vntt/vintt instructions (count = 1) and aloha_vp_ntt_rows.
aloha_status run_transforms(aloha_vp *vp, const std::vector<LimbJob> &jobs1, unsigned logn, bool inverse) {
  const uint32_t count = (uint32_t)jobs1.size();
  if (!count) return ALOHA_OK;
  // Plans (device job/tile arrays + chunk schedule) are cached: a steady-state caller that
  // repeats a batch pays only for the kernel launches.
  TransformPlan *plan = nullptr;
  uint64_t key_hash = 0xcbf29ce484222325ull;         // (one pass over the jobs instead of a memcmp against every cached array)
  for (uint32_t i = 0; i < count; i++) {
    const LimbJob &j = jobs1[i];
    for (uint64_t x : {(uint64_t)(uintptr_t)j.src, (uint64_t)(uintptr_t)j.dst, ((uint64_t)j.prime << 1) | j.scaled, (uint64_t)j.sc[0], (uint64_t)j.sc[2], (uint64_t)j.addc, ((uint64_t)j.zk << 32) | j.zkinv})
      key_hash = (key_hash ^ x) * 0x100000001b3ull;
  }
  for (auto &p : vp->plans)
    if (p->key_hash == key_hash && p->logn == logn && p->jobs.size() == 2 * (size_t)count &&
        memcmp(p->jobs.data(), jobs1.data(), count * sizeof(LimbJob)) == 0) { plan = p.get(); break; }
  if (!plan) {
    std::unique_ptr<TransformPlan> np(new TransformPlan());
    np->logn = logn;
    np->key_hash = key_hash;
    np->jobs = jobs1;
    np->jobs.resize(2 * (size_t)count);
    for (uint32_t i = 0; i < count; i++) { np->jobs[count + i] = jobs1[i]; np->jobs[count + i].src = jobs1[i].dst; }
    // Chunk schedule: both passes of a chunk run back to back so that the intermediate
    // (column-pass output) is still in the 256 MiB Infinity Cache when the row pass reads it.
    const uint32_t ch = vp->chunk_limbs ? vp->chunk_limbs : count;
    std::vector<TileDesc> tiles;
    bool tiled = !(vp->flags & ALOHA_FLAG_RADIX2_ONLY);
    for (uint32_t b = 0; b < count && tiled; b += ch) {
      const uint32_t n = std::min(ch, count - b);
      std::vector<TileDesc> uni, glb;
      aloha::k::build_row_tiles(np->jobs.data(), b, n, logn - 8, uni, glb);
      if (uni.empty() && glb.empty()) { tiled = false; break; }
      PlanChunk pc;
      pc.limb_begin = b; pc.limb_count = n;
      pc.uni_off = (uint32_t)tiles.size(); pc.n_uni = (uint32_t)uni.size();
      tiles.insert(tiles.end(), uni.begin(), uni.end());
      pc.glb_off = (uint32_t)tiles.size(); pc.n_glb = (uint32_t)glb.size();
      tiles.insert(tiles.end(), glb.begin(), glb.end());
      np->chunks.push_back(pc);
    }
    if (!tiled) np->chunks.clear();
#ifdef ALOHA_EXPERIMENTS
    // N = 2^16, one chunk, every row tile a bundle tile, a batch large enough to fill the persistent grid: the forward
    // transform runs as ONE launch with the intermediate handed over through the Infinity Cache (experiments/ntt_tile_knobs.hip)
    std::vector<uint64_t> tickets;
    if (logn == 16 && np->chunks.size() == 1 && np->chunks[0].n_glb == 0 && vp->fused_min && count >= vp->fused_min &&
        !(vp->flags & ALOHA_FLAG_NTT_TWO_PASS))
      aloha::k::build_fused_schedule(tiles, logn - 8, vp->fused_delay, tickets);
#endif
    HIPCK(vp, hipMalloc(&np->d_jobs, np->jobs.size() * sizeof(LimbJob)));
    HIPCK(vp, hipMemcpyAsync(np->d_jobs, np->jobs.data(), np->jobs.size() * sizeof(LimbJob), hipMemcpyHostToDevice, vp->stream));
    if (!tiles.empty()) {
      HIPCK(vp, hipMalloc(&np->d_tiles, tiles.size() * sizeof(TileDesc)));
      HIPCK(vp, hipMemcpyAsync(np->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, vp->stream));
    }
#ifdef ALOHA_EXPERIMENTS
    if (!tickets.empty()) {
      np->npos = (uint32_t)tickets.size();
      HIPCK(vp, hipMalloc(&np->d_tickets, tickets.size() * 8));
      HIPCK(vp, hipMemcpyAsync(np->d_tickets, tickets.data(), tickets.size() * 8, hipMemcpyHostToDevice, vp->stream));
      HIPCK(vp, hipMalloc(&np->d_ctl, aloha::k::fused_ctl_words(count) * 4));
    }
#endif
    HIPCK(vp, hipStreamSynchronize(vp->stream));    // pageable sources go out of scope
    if (vp->plans.size() >= kMaxPlans) {            // evict the oldest; its kernels have drained (sync above)
      vp->plans.erase(vp->plans.begin());
      vp->epoch++;                                  // recorded runs may point to it
    }
    plan = np.get();
    vp->plans.push_back(std::move(np));
  }
  const TransformPlan *pl = plan;
  bool any_scaled = false;
  for (uint32_t i = 0; i < count && inverse; i++) any_scaled = any_scaled || jobs1[i].scaled;
  const hipError_t e = issue(vp, [vp, pl, count, logn, inverse, any_scaled]() -> hipError_t {
    uint64_t launches = 0;
    hipError_t e = hipSuccess;
    if (pl->chunks.empty()) {
      e = aloha::k::ntt_radix2_batch(vp->stream, pl->d_jobs, vp->d_primes, count, logn, 0, logn, inverse, true, true, &launches);
#ifdef ALOHA_EXPERIMENTS
    } else if (!inverse && pl->d_tickets) {
      e = aloha::k::ntt_fused_fwd(vp->stream, pl->d_jobs, pl->d_jobs + count, vp->d_primes, pl->d_tiles + pl->chunks[0].uni_off,
                                  pl->d_tickets, pl->npos, count, pl->d_ctl, vp->d_fused_status, vp->fused_wgs, &launches);
      vp->fused_unchecked = true;
      vp->stats.fused_launches++;
      if (e == hipSuccess && vp->fused_stats) {        // debugging aid: synchronises
        uint32_t w[2] = {0, 0};
        if (hipMemcpyAsync(w, pl->d_ctl + aloha::k::kFusedStatsWord, 8, hipMemcpyDeviceToHost, vp->stream) == hipSuccess &&
            hipStreamSynchronize(vp->stream) == hipSuccess)
          fprintf(stderr, "aloha fused ntt: %u limbs, %u positions, %u tiles waited, %u poll iterations\n", count, pl->npos, w[0], w[1]);
      }
#endif
    } else {
      for (const PlanChunk &pc : pl->chunks) {
        e = aloha::k::ntt_tile_batch(vp->stream, pl->d_jobs, pl->d_jobs + count, vp->d_primes, pc.limb_begin,
                                     pc.limb_count, logn, inverse, pl->d_tiles + pc.uni_off, pc.n_uni,
                                     pl->d_tiles + pc.glb_off, pc.n_glb, &launches,
                                     (vp->flags & ALOHA_FLAG_NTT_TWO_PASS) ? 0u : (vp->flags & ALOHA_FLAG_NTT_SINGLE_14) ? 14u : 13u,
                                     any_scaled);
        if (e != hipSuccess) break;
      }
    }
    vp->stats.kernel_launches += launches;
    return e;
  });
  if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "transform launch failed: %s", hipGetErrorString(e));
  return ALOHA_OK;
}

// vle/vse address: addr_gen.v:44 (imm + cnt<<10) and the pointer select of
// vp_top_full.sv:105-117: row = imm[47:10] + PTR[imm[63:48]] with PTR 0 src0, 1 src1, 2 rslt,
// others 0; a LOAD with select 15 reads the KSK memory at csr_ksk_ptr (:113-114).  The RTL
// keeps 16 row bits; here the range is checked against the configured space instead.
u64 *mem_rows(aloha_vp *vp, uint64_t imm, bool is_load, const uint64_t csr[4], uint64_t nrows) {
  const unsigned sel = (unsigned)(imm >> 48);
  const uint64_t off = (imm & 0xFFFFFFFFFFFFull) >> 10;
  uint64_t row, rows;
  u64 *base;
  if (is_load && sel == 15) { row = off + csr[3]; rows = vp->ksk_rows; base = vp->d_ksk; }
  else { row = off + (sel == 0 ? csr[0] : sel == 1 ? csr[1] : sel == 2 ? csr[2] : 0); rows = vp->spm_rows; base = vp->d_spm; }
  if (row > rows || nrows > rows - row) return nullptr;
  return base + row * kLanes;
}

}  // namespace

// ============================================================================================
extern "C" {

uint32_t aloha_vp_abi_version(void) { return ALOHA_VP_ABI_VERSION; }

aloha_status aloha_vp_get_config(const aloha_vp *vp, aloha_vp_config *out) {
  if (!vp || !out) return ALOHA_ERR_ARG;
  out->abi_version = ALOHA_VP_ABI_VERSION; out->device = vp->device; out->spm_rows = vp->spm_rows; out->ksk_rows = vp->ksk_rows;
  out->max_log_n = vp->max_logn; out->flags = vp->flags; out->stream = (void *)vp->stream;
  out->iram_depth = vp->iram_depth; out->reserved = 0;
  return ALOHA_OK;
}

aloha_status aloha_vp_set_scratch(aloha_vp *vp, uint64_t row, uint64_t nrows) {
  if (!vp) return ALOHA_ERR_ARG;
  if (nrows && (row > vp->spm_rows || nrows > vp->spm_rows - row))
    return fail(vp, ALOHA_ERR_ADDR, "set_scratch: rows [%llu, +%llu) outside the SPM space", (u64)row, (u64)nrows);
  vp->scratch_row = nrows ? row : 0;
  vp->scratch_rows = nrows;
  return ALOHA_OK;
}

const char *aloha_status_string(aloha_status s) {
  switch (s) {
    case ALOHA_OK: return "ok";
    case ALOHA_ERR_ARG: return "bad argument";
    case ALOHA_ERR_VL: return "vector length not 64 * 2^k, k in [10,16]";
    case ALOHA_ERR_MODULUS: return "modulus not a prime in (2^59, 2^60)";
    case ALOHA_ERR_ROOT: return "modulus has no 2N-th root of unity";
    case ALOHA_ERR_DECODE: return "undefined instruction encoding";
    case ALOHA_ERR_BANK: return ".vv operands in the same register bank";
    case ALOHA_ERR_ADDR: return "memory row out of range";
    case ALOHA_ERR_AUT: return "even automorphism exponent";
    case ALOHA_ERR_NOBREAK: return "program ran off the instruction ROM";
    case ALOHA_ERR_DEVICE: return "HIP device error";
    case ALOHA_ERR_NOMEM: return "out of memory";
  }
  return "unknown";
}

const char *aloha_vp_last_error(const aloha_vp *vp) { return vp ? vp->err.c_str() : "null handle"; }

aloha_status aloha_vp_create(const aloha_vp_config *cfg, aloha_vp **out) {
  if (!cfg || !out) return ALOHA_ERR_ARG;
  *out = nullptr;
  if (cfg->abi_version != ALOHA_VP_ABI_VERSION) return ALOHA_ERR_ARG;
  if (cfg->max_log_n < 10 || cfg->max_log_n > 16) return ALOHA_ERR_ARG;
  if (cfg->iram_depth && (cfg->iram_depth < ALOHA_IRAM_DEPTH || cfg->iram_depth > ALOHA_IRAM_DEPTH_MAX)) return ALOHA_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return ALOHA_ERR_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return ALOHA_ERR_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ALOHA_ERR_DEVICE;   // kernels are gfx950 code objects
  aloha_vp *vp = new aloha_vp();
  vp->device = cfg->device;
  vp->flags = cfg->flags;
  vp->spm_rows = cfg->spm_rows;
  vp->ksk_rows = cfg->ksk_rows;
  vp->max_logn = cfg->max_log_n;
  vp->iram_depth = cfg->iram_depth ? cfg->iram_depth : ALOHA_IRAM_DEPTH;
  if (const char *e = getenv("ALOHA_NTT_CHUNK")) vp->chunk_limbs = (uint32_t)strtoul(e, nullptr, 10);
  vp->run_cache_on = !getenv("ALOHA_NO_RUN_CACHE");          // (A/B switch: every run goes through the sequencer again)
  vp->prod_disabled = getenv("ALOHA_PRODUCT_FALLBACK") != nullptr;   // (A/B and test switch: every polynomial product as five launches)
  vp->flow_on = !getenv("ALOHA_NO_FLOW");                    // (A/B switch: small-ring programs through the in-order sequencer, as in rounds 1-3)
  if (const char *e = getenv("ALOHA_MAX_TWIDDLE_SETS")) vp->max_slots = std::min(kMaxPrimes, std::max(1u, (uint32_t)strtoul(e, nullptr, 10)));
#ifdef ALOHA_EXPERIMENTS
  vp->graphs_on = getenv("ALOHA_RUN_GRAPH") != nullptr;
  // fused single-launch forward transform (N = 2^16), an experiment kept behind a switch: ALOHA_NTT_FUSED = the smallest
  // batch it serves (unset or 0: never);
  // ALOHA_NTT_FUSED_DELAY = limb groups between a column tile and the row tiles that read it;
  // ALOHA_NTT_FUSED_WGS = workgroups per CU (persistent; default and limit: what LDS and registers admit)
  if (const char *e = getenv("ALOHA_NTT_FUSED")) vp->fused_min = (uint32_t)strtoul(e, nullptr, 10);
  vp->fused_stats = getenv("ALOHA_NTT_FUSED_STATS") != nullptr;
  if (const char *e = getenv("ALOHA_NTT_FUSED_DELAY")) vp->fused_delay = std::max(1u, (uint32_t)strtoul(e, nullptr, 10));
  {
    uint32_t per_cu = aloha::k::fused_workgroups_per_cu();
    if (const char *e = getenv("ALOHA_NTT_FUSED_WGS")) per_cu = std::min(per_cu, std::max(1u, (uint32_t)strtoul(e, nullptr, 10)));
    vp->fused_wgs = (uint32_t)prop.multiProcessorCount * per_cu;
    // ALOHA_NTT_FUSED_MODE=persistent: persistent workgroups drawing tickets; default: one workgroup per ticket
    const char *m = getenv("ALOHA_NTT_FUSED_MODE");
    if (!(m && !strcmp(m, "persistent"))) vp->fused_wgs = 0;
  }
#endif
  vp->inst.assign(vp->iram_depth, 0);
  vp->imm.assign(vp->iram_depth, 0);
  auto bail = [&](const char *what) {
    fprintf(stderr, "aloha_vp_create: %s failed\n", what);
    aloha_vp_destroy(vp);
    return ALOHA_ERR_DEVICE;
  };
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev_device};   // the caller's device is current again on return
  if (hipSetDevice(vp->device) != hipSuccess) return bail("hipSetDevice");
  if (cfg->stream) vp->stream = (hipStream_t)cfg->stream;
  else {
    if (hipStreamCreateWithFlags(&vp->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    vp->own_stream = true;
  }
  const size_t vreg_bytes = (size_t)ALOHA_NVREG << (vp->max_logn + 3);
  if (hipMalloc(&vp->d_spm, (vp->spm_rows ? vp->spm_rows : 1) * (size_t)ALOHA_ROW_BYTES) != hipSuccess) return bail("hipMalloc(spm)");
  if (hipMalloc(&vp->d_ksk, (vp->ksk_rows ? vp->ksk_rows : 1) * (size_t)ALOHA_ROW_BYTES) != hipSuccess) return bail("hipMalloc(ksk)");
  if (hipMalloc(&vp->d_vreg, vreg_bytes) != hipSuccess) return bail("hipMalloc(vreg)");
  if (hipMalloc(&vp->d_scratch, (size_t)8 << vp->max_logn) != hipSuccess) return bail("hipMalloc(scratch)");
#ifdef ALOHA_TILE_TRACE
  if (hipMalloc(&vp->d_primes, kMaxPrimes * sizeof(PrimeDev) + aloha::k::kTileTraceWords * 8) != hipSuccess) return bail("hipMalloc(primes)");
#else
  if (hipMalloc(&vp->d_primes, kMaxPrimes * sizeof(PrimeDev)) != hipSuccess) return bail("hipMalloc(primes)");
#endif
#ifdef ALOHA_EXPERIMENTS
  if (hipMalloc(&vp->d_fused_status, 4) != hipSuccess) return bail("hipMalloc(status)");
  if (hipMemsetAsync(vp->d_fused_status, 0, 4, vp->stream) != hipSuccess) return bail("memset");
#endif
  // the RTL's memories power up as X; give them a defined 0 instead
  if (hipMemsetAsync(vp->d_spm, 0, (vp->spm_rows ? vp->spm_rows : 1) * (size_t)ALOHA_ROW_BYTES, vp->stream) != hipSuccess) return bail("memset");
  if (hipMemsetAsync(vp->d_ksk, 0, (vp->ksk_rows ? vp->ksk_rows : 1) * (size_t)ALOHA_ROW_BYTES, vp->stream) != hipSuccess) return bail("memset");
  if (hipMemsetAsync(vp->d_vreg, 0, vreg_bytes, vp->stream) != hipSuccess) return bail("memset");
  if (hipStreamSynchronize(vp->stream) != hipSuccess) return bail("sync");
  *out = vp;
  return ALOHA_OK;
}

void aloha_vp_destroy(aloha_vp *vp) {
  if (!vp) return;
  DeviceGuard dev_guard(vp);
  if (vp->stream) (void)hipStreamSynchronize(vp->stream);
  for (auto &s : vp->slots) if (s.d_tables) (void)hipFree(s.d_tables);
  if (vp->d_spm) (void)hipFree(vp->d_spm);
  if (vp->d_ksk) (void)hipFree(vp->d_ksk);
  if (vp->d_vreg) (void)hipFree(vp->d_vreg);
  if (vp->d_scratch) (void)hipFree(vp->d_scratch);
  if (vp->d_primes) (void)hipFree(vp->d_primes);
  vp->plans.clear();
  for (auto &b : vp->blobs) if (b.second) (void)hipFree(b.second);
  for (auto &e : vp->enc_tables) if (e.second.alloc) (void)hipFree(e.second.alloc);
#ifdef ALOHA_EXPERIMENTS
  if (vp->d_fused_status) (void)hipFree(vp->d_fused_status);
#endif
  if (vp->d_enc_in) (void)hipFree(vp->d_enc_in);
  if (vp->d_enc_work) (void)hipFree(vp->d_enc_work);
  if (vp->d_prod_work) (void)hipFree(vp->d_prod_work);
  if (vp->d_prod_fb) (void)hipFree(vp->d_prod_fb);
  if (vp->d_prod_ctl) (void)hipFree(vp->d_prod_ctl);
  if (vp->d_flow) (void)hipFree(vp->d_flow);
  prod_release(vp);
  if (vp->own_stream && vp->stream) (void)hipStreamDestroy(vp->stream);
  delete vp;
}

static u64 *space_ptr(aloha_vp *vp, int space, uint64_t row, uint64_t bytes) {
  if (space != ALOHA_SPACE_SPM && space != ALOHA_SPACE_KSK) return nullptr;
  const uint64_t rows = space == ALOHA_SPACE_SPM ? vp->spm_rows : vp->ksk_rows;
  if (row > rows || bytes > (rows - row) * (uint64_t)ALOHA_ROW_BYTES) return nullptr;
  return (space == ALOHA_SPACE_SPM ? vp->d_spm : vp->d_ksk) + row * kLanes;
}

aloha_status aloha_vp_write(aloha_vp *vp, int space, uint64_t row, const void *src, uint64_t bytes) {
  if (!vp || !src) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  u64 *p = space_ptr(vp, space, row, bytes);
  if (!p) return fail(vp, ALOHA_ERR_ADDR, "write: rows [%llu, +%llu B) outside space %d", (u64)row, (u64)bytes, space);
  HIPCK(vp, hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));      // host pointer is borrowed for the call only
  return ALOHA_OK;
}

aloha_status aloha_vp_read(aloha_vp *vp, int space, uint64_t row, void *dst, uint64_t bytes) {
  if (!vp || !dst) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  u64 *p = space_ptr(vp, space, row, bytes);
  if (!p) return fail(vp, ALOHA_ERR_ADDR, "read: rows [%llu, +%llu B) outside space %d", (u64)row, (u64)bytes, space);
  HIPCK(vp, hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return check_fused(vp);
}

void *aloha_vp_device_ptr(aloha_vp *vp, int space, uint64_t row) {
  return vp ? space_ptr(vp, space, row, 0) : nullptr;
}

void *aloha_vp_stream(aloha_vp *vp) { return vp ? (void *)vp->stream : nullptr; }
int aloha_vp_device(const aloha_vp *vp) { return vp ? vp->device : -1; }

aloha_status aloha_vp_load_isram(aloha_vp *vp, uint32_t base_pc, const uint32_t *words, uint32_t n) {
  if (!vp || !words) return ALOHA_ERR_ARG;
  if (base_pc > vp->iram_depth || n > vp->iram_depth - base_pc) return fail(vp, ALOHA_ERR_ADDR, "isram [%u, +%u) outside the %u-word ROM", base_pc, n, vp->iram_depth);
  for (uint32_t i = 0; i < n; i++) {
    vp->inst[base_pc + i] = words[3 * i];
    vp->imm[base_pc + i] = ((uint64_t)words[3 * i + 1] << 32) | words[3 * i + 2];
  }
  return ALOHA_OK;
}

aloha_status aloha_vp_load_isram_hex(aloha_vp *vp, uint32_t base_pc, const char *text) {
  if (!vp || !text) return ALOHA_ERR_ARG;
  // $readmemh subset: whitespace-separated hex words of up to 24 digits, // comments, '_' separators
  std::vector<uint32_t> words;
  const char *p = text;
  while (*p) {
    if (p[0] == '/' && p[1] == '/') { while (*p && *p != '\n') p++; continue; }
    if (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') { p++; continue; }
    unsigned __int128 v = 0;
    int digits = 0;
    while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') {
      const char c = *p++;
      if (c == '_') continue;
      int d;
      if (c >= '0' && c <= '9') d = c - '0';
      else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
      else return fail(vp, ALOHA_ERR_ARG, "isram text: bad character '%c'", c);
      if (++digits > 24) return fail(vp, ALOHA_ERR_ARG, "isram text: word wider than 96 bit");
      v = (v << 4) | (unsigned)d;
    }
    words.push_back((uint32_t)(v >> 64));
    words.push_back((uint32_t)(v >> 32));
    words.push_back((uint32_t)v);
  }
  return aloha_vp_load_isram(vp, base_pc, words.data(), (uint32_t)(words.size() / 3));
}

#ifdef ALOHA_EXPERIMENTS
// variant builds with -DALOHA_PROD_TRACE: the clock stamps the product kernel left behind its scratch (tests/perf/product_trace.py)
aloha_status aloha_vp_debug_product_trace(aloha_vp *vp, void *dst, uint64_t bytes) {
  if (!vp || !dst || !vp->d_prod_work || bytes > (1u << 20)) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  HIPCK(vp, hipMemcpyAsync(dst, vp->d_prod_work + aloha::k::poly_product_work_words(), bytes, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return ALOHA_OK;
}
#endif

#ifdef ALOHA_TILE_TRACE
// variant builds with -DALOHA_TILE_TRACE: the clock stamps the tile kernels left behind the prime table (tests/perf/tile_trace.py)
aloha_status aloha_vp_debug_tile_trace(aloha_vp *vp, void *dst, uint64_t bytes) {
  if (!vp || !dst || bytes > aloha::k::kTileTraceWords * 8ull) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  HIPCK(vp, hipMemcpyAsync(dst, vp->d_primes + kMaxPrimes, bytes, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return ALOHA_OK;
}
#endif

// (Round 4: polling hipStreamQuery for up to 60 us before the blocking call was measured and dropped -- hipStreamSynchronize already
// spins on the completion signal: one N = 2^12 transform with its synchronisation 18.2 us polled against 16.9 us, tests/perf/sync_latency.py.)
aloha_status aloha_vp_sync(aloha_vp *vp) {
  if (!vp) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return check_fused(vp);
}

}  // extern "C" (reopened after the sequencer)

// ---- the sequencer ---------------------------------------------------------------------------
// Plain sequential semantics (one instruction in flight, seq_top.v:343-413) with three exact
// program-level fusions, all disabled by ALOHA_FLAG_NO_FUSION:
//   * vle is zero-copy: the register ALIASES the memory rows until either is overwritten;
//   * `op vd, ...; vse vd, Z` with vd dead afterwards writes straight to Z;
//   * runs of `vle vA, X; vntt|vintt vB, vA; vse vB, Y` (any vsetq in between, vA/vB dead
//     afterwards) are collected into ONE batched tile transform over all their limbs.
// Vector registers are not observable after `break`, memory is: every byte a vse
// writes is identical to the unfused execution, in program order.
namespace {

struct Decoded {
  unsigned funct6, vs2, vs1, funct3, vd;
  uint64_t imm;
};

inline Decoded decode_at(const aloha_vp *vp, uint32_t pc) {                // expander.v:123-130
  const uint32_t inst = vp->inst[pc];
  return Decoded{inst >> 26, (inst >> 20) & 31, (inst >> 15) & 31, (inst >> 12) & 7, (inst >> 7) & 31, vp->imm[pc]};
}

bool reads_reg(const Decoded &d, unsigned r) {
  switch (d.funct6) {
    case 0x01: case 0x05: case 0x09:
      if (d.funct3 == 0) return d.vs1 == r || d.vs2 == r;
      if (d.funct3 == 1) return d.vs1 == r;
      if (d.funct3 == 2) return d.vs2 == r;
      return false;
    case 0x0d: case 0x11: case 0x15: case 0x19: case 0x1d: case 0x21: case 0x02: case 0x06: case 0x07: return d.vs1 == r;
    default: return false;
  }
}

bool writes_reg(const Decoded &d, unsigned r) {
  switch (d.funct6) {
    case 0x01: case 0x05: case 0x09: case 0x0d: case 0x11: case 0x15: case 0x19: case 0x1d: case 0x21: case 0x02: case 0x06: case 0x03:
      return d.vd == r;
    default: return false;
  }
}

// is register r dead from pc on (overwritten or program end before any read)?
bool dead_from(const aloha_vp *vp, uint32_t pc, unsigned r) {
  for (; pc < vp->iram_depth; pc++) {
    const Decoded d = decode_at(vp, pc);
    if (d.funct6 == 0x10) return true;
    if (reads_reg(d, r)) return false;
    if (writes_reg(d, r)) return true;
  }
  return true;
}

inline bool overlap(const u64 *a, const u64 *b, size_t n) { return a < b + n && b < a + n; }

// Device copy of a job array, cached by content: programs repeat, so in steady state a batched
// launch needs no upload.  Entries are immutable; eviction happens only after a stream sync.
aloha_status device_blob(aloha_vp *vp, const void *host, size_t bytes, const void **dev) {
  for (size_t i = 0; i < vp->blobs.size(); i++) {
    auto &b = vp->blobs[i];
    if (b.first.size() == bytes && memcmp(b.first.data(), host, bytes) == 0) {
      *dev = b.second;
      // least recently USED goes first: a flush hands out several blobs before it launches with them, and a miss among them evicts
      // the front entry -- which must not be one handed out a moment ago (a hit used to leave the entry where it was)
      if (i + 1 != vp->blobs.size()) std::rotate(vp->blobs.begin() + i, vp->blobs.begin() + i + 1, vp->blobs.end());
      return ALOHA_OK;
    }
  }
  void *d = nullptr;
  HIPCK(vp, hipMalloc(&d, bytes));
  HIPCK(vp, hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  if (vp->blobs.size() >= 512) {                             // everything queued has drained (sync above)
    (void)hipFree(vp->blobs.front().second);
    vp->blobs.erase(vp->blobs.begin());
    vp->epoch++;                                             // recorded runs may point to it
  }
  vp->blobs.emplace_back(std::vector<uint8_t>((const uint8_t *)host, (const uint8_t *)host + bytes), d);
  *dev = d;
  return ALOHA_OK;
}

// ---- dataflow execution of small-ring programs ----------------------------------------------------------------------------------
// The in-order sequencer below batches what stands TOGETHER in a program: runs of `vle; op; vse` over memory operands.  The
// reference's own programs are not written that way: keyswitch.mem:1-122 (N = 8192, two ciphertext primes and one special prime)
// keeps its values in registers and interleaves the work of its limbs -- 18 transforms, 4 automorphisms, 38 element-wise
// instructions, each a launch of its own on one 64 KiB limb: 60 launches of ~5 us, 307 us on the GPU against ~206 us derived
// for the RTL (measured in round 4, bench.py `reference_programs_n8192`).  At such sizes a launch costs
// its latency, not its work, so what a program costs is the DEPTH of its dependency graph, not its length.  For rings up to
// N = 2^13 a program segment is therefore first turned into that graph:
//   * registers are renamed (every write makes a new value with storage of its own: a pool buffer, or -- for a value that a
//     `vse` stores and nothing conflicts with -- the stored rows themselves), `vle` aliases its rows, so only TRUE dependencies
//     remain besides the memory hazards between loads and stores, which become edges too;
//   * every operation gets the level 1 + max(level of what it depends on); the operations of one level are independent, and
//     all of one kind leave as ONE batched launch (the same batched kernels the in-order sequencer uses);
//   * operations whose results nobody reads are dropped (registers are not observable after `break`).
// Every byte a `vse` writes is what sequential execution writes (the differential fuzz of tests/test_gpu_parity.py runs
// random programs through this path, the in-order engine, the unfused engine and the oracle).  Anything irregular -- an
// instruction that would fail, a vector-length change in mid-program, a ring above 2^13 -- is left to the in-order sequencer,
// which reports errors with their program counter: planning has no side effects, `handled` stays false.
struct FlowRange { const u64 *lo, *hi; };
inline bool flow_hit(const FlowRange &a, const FlowRange &b) { return a.lo < b.hi && b.lo < a.hi; }

aloha_status run_flow(aloha_vp *vp, uint32_t pc, const uint64_t csr[4], uint64_t step, bool *handled) {
  *handled = false;
  enum Kind { K_NTT, K_INTT, K_EW, K_AUT, K_COPY, K_SUM };
  struct Val {
    const u64 *fixed = nullptr;     // storage that exists already: rows a vle aliases, a vector register's slab, rows a vse targets
    bool alias = false;             // fixed storage is SPM / KSK memory (a loaded value)
    int producer = -1;              // operation that writes it (-1: exists on entry)
    int buf = -1;                   // pool buffer, assigned after the levels are known
    int def_level = 0, last_level = 0;
  };
  struct Op {
    Kind kind;
    int dst = -1, a = -1, b = -1;   // value ids
    uint64_t q = 0, scalar = 0;
    uint32_t prime = 0;
    int ewop = 0;
    uint32_t k = 0, kinv = 0, roll = 0;
    std::vector<int> pred;
    int level = 0;
    bool live = false;
    // K_SUM: a tree of element-wise instructions under one modulus collapsed into  sum_t x_t * y_t + sum_t x_t * s_t + cst
    std::vector<std::pair<int, int>> vv;           // products of two values
    std::vector<std::pair<int, uint64_t>> lin;     // value times scalar
    uint64_t cst = 0;
    int into = -1;                                 // absorbed by that (later) operation
    // K_INTT: `vintt ; vfqmul.vs c ; vfqadd.vs a` as ONE transform job (LimbJob::sc / addc: the scalar rides in the last stage)
    bool scaled = false;
    uint64_t mulc = 1, addc = 0;
  };
  struct MemAcc { FlowRange r; int op; bool write; };
  std::vector<Val> vals;
  std::vector<Op> ops;
  std::vector<MemAcc> acc;                                     // SPM / KSK accesses of the planned operations, in program order
  int regs[ALOHA_NVREG];
  for (unsigned r = 0; r < ALOHA_NVREG; r++) { Val v; v.fixed = vreg(vp, r); regs[r] = (int)vals.size(); vals.push_back(v); }
  uint64_t vl = vp->vl_bits, q = vp->q, mu = vp->mu;
  unsigned logn = 0;
  uint32_t n = 0;
  bool sized = false, touched = false;
  uint64_t ninstr = 0;
  auto resize = [&]() -> bool {
    if (vl_logn(vl, vp->max_logn, &logn) != ALOHA_OK || logn > 13) return false;
    n = 1u << logn; sized = true;
    return true;
  };
  auto new_op = [&](Kind k) -> int { Op o; o.kind = k; o.q = q; ops.push_back(o); return (int)ops.size() - 1; };
  auto depend = [&](int op, int on) { if (on >= 0 && on != op) ops[op].pred.push_back(on); };
  auto in_memory = [&](int v) {                                // the value's storage is SPM / KSK rows (loaded from them, or placed there by a vse)
    const u64 *f = vals[v].fixed;
    return f && !(f >= vp->d_vreg && f < vp->d_vreg + ((size_t)ALOHA_NVREG << vp->max_logn));
  };
  // operation `op` reads value `v`: a true dependency on its producer; a read of memory if the value lives in rows
  auto use = [&](int op, int v) {
    depend(op, vals[v].producer);
    if (in_memory(v)) {
      const FlowRange r{vals[v].fixed, vals[v].fixed + n};
      for (const MemAcc &m : acc) if (m.write && flow_hit(m.r, r)) depend(op, m.op);      // written earlier in the program
      acc.push_back(MemAcc{r, op, false});
    }
  };
  auto fresh = [&](int op) -> int { Val v; v.producer = op; vals.push_back(v); ops[op].dst = (int)vals.size() - 1; return ops[op].dst; };
  uint32_t end_pc = pc;
  for (uint32_t i = pc;; i++) {
    if (i >= vp->iram_depth || ops.size() > 4096) return ALOHA_OK;                           // (no break / too long: not handled)
    const Decoded d = decode_at(vp, i);
    ninstr++;
    const unsigned f6 = d.funct6;
    if (f6 == 0x10) { end_pc = i; break; }
    if (f6 == 0x00) continue;
    if (f6 == 0x04) { if (touched && d.imm != vl) return ALOHA_OK; vl = d.imm; if (!resize()) return ALOHA_OK; continue; }
    if (f6 == 0x08) { if (!modulus_ok(d.imm)) return ALOHA_OK; q = d.imm; continue; }
    if (f6 == 0x0c) { mu = d.imm; continue; }
    if (!sized && !resize()) return ALOHA_OK;
    touched = true;
    const bool needs_q = f6 != 0x03 && f6 != 0x07 && f6 != 0x19 && f6 != 0x1d;
    if (needs_q && !(q > (1ull << 59) && q < (1ull << 60))) return ALOHA_OK;
    switch (f6) {
      case 0x03: {                                                 // vle: the register aliases the rows
        const u64 *p = mem_rows(vp, d.imm, true, csr, n / kLanes);
        if (!p) return ALOHA_OK;
        Val v; v.fixed = p; v.alias = true;
        regs[d.vd] = (int)vals.size(); vals.push_back(v);
        break;
      }
      case 0x07: {                                                 // vse
        u64 *p = mem_rows(vp, d.imm, false, csr, n / kLanes);
        if (!p) return ALOHA_OK;
        const FlowRange r{p, p + n};
        const int v = regs[d.vs1];
        if (in_memory(v)) {
          if (vals[v].fixed == p) break;                          // rows stored onto themselves
          if (flow_hit(FlowRange{vals[v].fixed, vals[v].fixed + n}, r)) return ALOHA_OK;   // a shifted self-overlap: left to the in-order sequencer
        }
        // registers whose value lives in rows this store overwrites keep the OLD contents: copied out first
        for (unsigned rr = 0; rr < ALOHA_NVREG; rr++) {
          const int a = regs[rr];
          if (a == v || !in_memory(a) || !flow_hit(FlowRange{vals[a].fixed, vals[a].fixed + n}, r)) continue;
          const int c = new_op(K_COPY);
          ops[c].a = a; use(c, a);
          fresh(c);
          for (unsigned r2 = 0; r2 < ALOHA_NVREG; r2++) if (regs[r2] == a) regs[r2] = ops[c].dst;
        }
        // the producing operation may write the rows directly when nothing else planned so far touches them (an operation
        // that reads exactly these rows as its own operand may: element-wise and transform kernels run in place)
        const int pr = vals[v].producer;
        bool direct = pr >= 0 && !vals[v].fixed && ops[pr].kind != K_COPY;
        if (direct)
          for (const MemAcc &m : acc) {
            if (!flow_hit(m.r, r)) continue;
            const bool own_operand = m.op == pr && !m.write && m.r.lo == r.lo && ops[pr].kind != K_AUT;
            if (!own_operand) { direct = false; break; }
          }
        if (direct) {
          vals[v].fixed = p;
          acc.push_back(MemAcc{r, pr, true});
          for (size_t o2 = (size_t)pr + 1; o2 < ops.size(); o2++)   // operations that read the value now read these rows
            if (ops[o2].a == v || ops[o2].b == v) acc.push_back(MemAcc{r, (int)o2, false});
        } else {
          const int c = new_op(K_COPY);
          ops[c].a = v; use(c, v);
          for (const MemAcc &m : acc) if (flow_hit(m.r, r)) depend(c, m.op);            // after every earlier reader and writer of the rows
          Val t; t.fixed = p; t.producer = c; vals.push_back(t); ops[c].dst = (int)vals.size() - 1;
          acc.push_back(MemAcc{r, c, true});
        }
        break;
      }
      case 0x01: case 0x05: case 0x09: case 0x0d: case 0x11: {   // element-wise (the in-order sequencer's decode, expander.v:178-417)
        aloha::k::EwOp op;
        unsigned ra, rb = ALOHA_NVREG;
        uint64_t s = 0;
        if (f6 == 0x0d || f6 == 0x11) { ra = d.vs1; op = f6 == 0x0d ? aloha::k::EW_MOD : aloha::k::EW_CP; }
        else if (d.funct3 == 0) {
          if (((d.vs1 ^ d.vs2) & 1) == 0) return ALOHA_OK;
          ra = d.vs1; rb = d.vs2;
          op = f6 == 0x01 ? aloha::k::EW_MUL_VV : f6 == 0x05 ? aloha::k::EW_ADD_VV : aloha::k::EW_SUB_VV;
        } else if (d.funct3 == 1) {
          ra = d.vs1; s = d.imm;
          op = f6 == 0x01 ? aloha::k::EW_MUL_VS : f6 == 0x05 ? aloha::k::EW_ADD_VS : aloha::k::EW_SUB_VS;
        } else if (d.funct3 == 2 && f6 == 0x09) { ra = d.vs2; s = d.imm; op = aloha::k::EW_SUB_SV; }
        else return ALOHA_OK;
        s = s >= q ? s - q : s;
        if (s >= q) return ALOHA_OK;
        const int o = new_op(K_EW);
        ops[o].ewop = (int)op; ops[o].scalar = s;
        ops[o].a = regs[ra]; use(o, regs[ra]);
        if (rb < ALOHA_NVREG) { ops[o].b = regs[rb]; use(o, regs[rb]); }
        regs[d.vd] = fresh(o);
        break;
      }
      case 0x15: case 0x19: case 0x1d: case 0x21: {                // vaut / vroli / vautn / vautz
        if (f6 != 0x19 && !((step + d.imm) & 1)) return ALOHA_OK;
        const uint64_t k = (step + d.imm) & (n - 1);
        const int o = new_op(K_AUT);
        ops[o].k = (uint32_t)k;
        ops[o].roll = f6 == 0x19 ? 1 : f6 == 0x1d ? 2 : f6 == 0x21 ? 3 : 0;
        if (ops[o].roll == 0 || ops[o].roll == 3) {
          uint64_t kinv = 1;
          for (int it = 0; it < 5; it++) kinv = kinv * (2 - k * kinv);
          ops[o].kinv = (uint32_t)(kinv & (n - 1));
        }
        ops[o].a = regs[d.vs1]; use(o, regs[d.vs1]);
        regs[d.vd] = fresh(o);
        break;
      }
      case 0x02: case 0x06: {                                      // vntt / vintt
        if ((q - 1) % (2ull << logn)) return ALOHA_OK;
        int src = regs[d.vs1];
        // `vcp t, x ; vntt vd, t` (keyswitch.mem:9-10,:13-14,:29-30,:33-34: a limb reinterpreted under another modulus): the forward
        // transform takes any word < 2^60 < 2q and reduces as it goes, and vcp / vfqmod under the SAME modulus only subtract a
        // multiple of q -- the transform of x is the transform of t, word for word, one level earlier
        if (f6 == 0x02 && vals[src].producer >= 0) {
          const Op &c = ops[vals[src].producer];
          if (c.kind == K_EW && (c.ewop == (int)aloha::k::EW_CP || c.ewop == (int)aloha::k::EW_MOD) && c.q == q && !in_memory(c.a)) src = c.a;
        }
        const int o = new_op(f6 == 0x06 ? K_INTT : K_NTT);
        ops[o].a = src; use(o, src);
        regs[d.vd] = fresh(o);
        break;
      }
      default: return ALOHA_OK;
    }
  }
  (void)end_pc;
  // ---- trees of element-wise instructions as ONE pass each --------------------------------------------------------------------
  // `vfqmul.vv t, a, b ; vfqadd.vv acc, acc, t` (keyswitch.mem:46-79) or `vfqsub.vv d, acc, t ; vfqmul.vs e, d, c ; vfqadd.vv r, r, e`
  // (:102-117) are exact arithmetic mod q on canonical residues: a tree of such instructions under one modulus whose intermediate
  // values have no other reader equals  sum x y + sum x s + cst  (mod q) over its leaves, which the sums' kernel evaluates in one
  // pass (dot_batch: the in-order sequencer's lazy sums, there for memory operands only).  One level instead of the tree's depth.
  // A leaf read moves to the time of the tree's root: refused when rows it reads are written in between.
  {
    std::vector<int> uses(vals.size(), 0);
    for (const Op &o : ops) for (int v : {o.a, o.b}) if (v >= 0) uses[v]++;
    auto mem_written_between = [&](int v, int from, int to) {      // is storage of value v (rows) written by an operation in (from, to)?
      if (!in_memory(v)) return false;
      const FlowRange r{vals[v].fixed, vals[v].fixed + n};
      for (int w = from + 1; w < to; w++)
        if (ops[w].dst >= 0 && in_memory(ops[w].dst) && flow_hit(FlowRange{vals[ops[w].dst].fixed, vals[ops[w].dst].fixed + n}, r)) return true;
      return false;
    };
    // `vintt t, x ; [vfqmul.vs u, t, c ;] [vfqadd.vs | vfqsub.vs w, u, a]` (keyswitch.mem:80-83): the scalar rides in the transform's
    // last stage, the constant on its canonical result -- the in-order sequencer's scaled transform job, here for register operands
    if (logn >= 12)
      for (int i = 0; i < (int)ops.size(); i++) {
        Op &o = ops[i];
        if (o.kind != K_EW || o.a < 0) continue;
        const bool is_mul = o.ewop == (int)aloha::k::EW_MUL_VS, is_add = o.ewop == (int)aloha::k::EW_ADD_VS, is_sub = o.ewop == (int)aloha::k::EW_SUB_VS;
        if (!is_mul && !is_add && !is_sub) continue;
        const int v = o.a, p = vals[v].producer;
        if (p < 0 || ops[p].kind != K_INTT || ops[p].q != o.q || uses[v] != 1 || vals[v].fixed || ops[p].addc) continue;
        if (is_mul && ops[p].scaled) continue;
        if (mem_written_between(ops[p].a, p, i)) continue;
        const uint64_t c = ops[p].mulc, a0 = is_add ? o.scalar : is_sub ? (o.scalar ? o.q - o.scalar : 0) : 0;
        o.kind = K_INTT; o.a = ops[p].a; o.b = -1; o.scaled = true;
        o.mulc = is_mul ? o.scalar : c; o.addc = a0;
        o.pred.insert(o.pred.end(), ops[p].pred.begin(), ops[p].pred.end());
        ops[p].into = i;
        uses[v] = 0;
      }
    struct Aff { std::vector<std::pair<int, int>> vv; std::vector<std::pair<int, uint64_t>> lin; uint64_t cst = 0; std::vector<int> absorbed; };
    std::vector<Aff> aff(ops.size());
    std::vector<char> has(ops.size(), 0);
    // the sole reader of an instruction's result, when that is an element-wise instruction: a form that a scalar will multiply
    // (vfqmul.vs, vfqsub.sv) must stay free of vector products, so it does not absorb them (they stay a sum of their own, a level
    // earlier and usually off the critical path: (acc - t) * P^-1 + base is then ONE pass, keyswitch.mem:102-117)
    std::vector<int> reader(ops.size(), -1);
    for (int i = 0; i < (int)ops.size(); i++)
      for (int v : {ops[i].a, ops[i].b})
        if (v >= 0 && vals[v].producer >= 0 && uses[v] == 1) reader[vals[v].producer] = i;
    auto scalar_reader = [&](int i) {
      const int r = reader[i];
      return r >= 0 && ops[r].kind == K_EW && ops[r].q == ops[i].q && (ops[r].ewop == (int)aloha::k::EW_MUL_VS || ops[r].ewop == (int)aloha::k::EW_SUB_SV);
    };
    for (int i = 0; i < (int)ops.size(); i++) {
      Op &o = ops[i];
      if (o.kind != K_EW) continue;
      const uint64_t qq = o.q;
      auto neg = [&](uint64_t x) { return x ? qq - x : 0; };
      // an operand as an affine form: the tree below it if it may be absorbed, else the value itself as a leaf
      const bool no_vv = scalar_reader(i);
      auto operand = [&](int v, bool allow_vv) -> Aff {
        const int p = vals[v].producer;
        if (no_vv) allow_vv = false;
        if (p >= 0 && ops[p].into < 0 && has[p] && ops[p].q == qq && uses[v] == 1 && !vals[v].fixed && (allow_vv || aff[p].vv.empty())) {
          bool ok = true;
          std::vector<int> all(aff[p].absorbed); all.push_back(p);
          for (int a : all) for (int x : {ops[a].a, ops[a].b}) if (x >= 0 && mem_written_between(x, a, i)) ok = false;
          if (ok) { Aff f = aff[p]; f.absorbed.push_back(p); return f; }
        }
        Aff f; f.lin.emplace_back(v, 1); return f;
      };
      Aff r;
      bool ok = true;
      switch ((aloha::k::EwOp)o.ewop) {
        case aloha::k::EW_MUL_VV: r.vv.emplace_back(o.a, o.b); break;
        case aloha::k::EW_CP: case aloha::k::EW_MOD: r = operand(o.a, true); break;
        case aloha::k::EW_ADD_VV: case aloha::k::EW_SUB_VV: {
          r = operand(o.a, true);
          Aff y = operand(o.b, o.ewop == (int)aloha::k::EW_ADD_VV);
          if (o.ewop == (int)aloha::k::EW_SUB_VV) { for (auto &t : y.lin) t.second = neg(t.second); y.cst = neg(y.cst); }
          r.vv.insert(r.vv.end(), y.vv.begin(), y.vv.end());
          r.lin.insert(r.lin.end(), y.lin.begin(), y.lin.end());
          r.cst = (r.cst + y.cst) % qq;
          r.absorbed.insert(r.absorbed.end(), y.absorbed.begin(), y.absorbed.end());
          break;
        }
        case aloha::k::EW_MUL_VS: {
          r = operand(o.a, false);
          for (auto &t : r.lin) t.second = aloha::nt::mulmod(t.second, o.scalar, qq);
          r.cst = aloha::nt::mulmod(r.cst, o.scalar, qq);
          break;
        }
        case aloha::k::EW_ADD_VS: r = operand(o.a, true); r.cst = (r.cst + o.scalar) % qq; break;
        case aloha::k::EW_SUB_VS: r = operand(o.a, true); r.cst = (r.cst + neg(o.scalar)) % qq; break;
        case aloha::k::EW_SUB_SV: {
          r = operand(o.a, false);
          for (auto &t : r.lin) t.second = neg(t.second);
          r.cst = (neg(r.cst) + o.scalar) % qq;
          break;
        }
        default: ok = false;
      }
      if (!ok || r.vv.size() + r.lin.size() + 1 > (size_t)aloha::k::kDotTerms) continue;     // (stays a plain instruction, a leaf for its readers)
      aff[i] = r; has[i] = 1;
    }
    // roots: forms that absorbed something and are not absorbed themselves
    std::vector<char> gone(ops.size(), 0);
    for (int i = (int)ops.size() - 1; i >= 0; i--) {
      if (!has[i] || gone[i] || aff[i].absorbed.empty()) continue;
      Op &o = ops[i];
      o.kind = K_SUM; o.vv = aff[i].vv; o.lin = aff[i].lin; o.cst = aff[i].cst;
      for (int a : aff[i].absorbed) {
        gone[a] = 1; ops[a].into = i;
        o.pred.insert(o.pred.end(), ops[a].pred.begin(), ops[a].pred.end());
      }
    }
    for (Op &o : ops)                                               // edges to an absorbed instruction now lead to the sum that holds it
      for (int &p2 : o.pred) { while (ops[p2].into >= 0) p2 = ops[p2].into; }
    for (int i = 0; i < (int)ops.size(); i++) {
      Op &o = ops[i];
      o.pred.erase(std::remove(o.pred.begin(), o.pred.end(), i), o.pred.end());
    }
  }
  // ---- what is needed: operations that write memory, and what they depend on ----------------------------------------------
  for (size_t i = ops.size(); i-- > 0;) {
    Op &o = ops[i];
    if (o.dst >= 0 && vals[o.dst].fixed && vals[o.dst].producer == (int)i && !vals[o.dst].alias &&
        !(vals[o.dst].fixed >= vp->d_vreg && vals[o.dst].fixed < vp->d_vreg + ((size_t)ALOHA_NVREG << vp->max_logn)))
      o.live = true;                                               // its destination is memory rows
    if (o.into >= 0) o.live = false;                               // (its arithmetic happens inside the sum that absorbed it)
    if (!o.live) continue;
    for (int p : o.pred) ops[p].live = true;
  }
  auto operands_of = [&](const Op &o, std::vector<int> &out) {
    out.clear();
    if (o.kind == K_SUM) { for (const auto &t : o.vv) { out.push_back(t.first); out.push_back(t.second); } for (const auto &t : o.lin) out.push_back(t.first); }
    else { if (o.a >= 0) out.push_back(o.a); if (o.b >= 0) out.push_back(o.b); }
  };
  std::vector<int> opnds;
  size_t nlive = 0;
  int max_level = 0;
  for (size_t i = 0; i < ops.size(); i++) {
    Op &o = ops[i];
    if (!o.live) continue;
    nlive++;
    int lv = 0;
    for (int p : o.pred) if (ops[p].live) lv = std::max(lv, ops[p].level);
    o.level = lv + 1;
    max_level = std::max(max_level, o.level);
    if (o.dst >= 0) { vals[o.dst].def_level = o.level; vals[o.dst].last_level = std::max(vals[o.dst].last_level, o.level); }
    operands_of(o, opnds);
    for (int v : opnds) vals[v].last_level = std::max(vals[v].last_level, o.level);
  }
  // ---- pool buffers for the values without storage of their own: an interval colouring over the levels ---------------------
  std::vector<int> order;                                           // live values that need a buffer, by definition level
  for (size_t v = 0; v < vals.size(); v++)
    if (!vals[v].fixed && vals[v].producer >= 0 && ops[vals[v].producer].live) order.push_back((int)v);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return vals[x].def_level < vals[y].def_level; });
  std::vector<int> free_at;                                         // per buffer: the last level at which it is read
  for (int v : order) {
    int b = -1;
    for (size_t k = 0; k < free_at.size(); k++) if (free_at[k] < vals[v].def_level) { b = (int)k; break; }
    if (b < 0) { b = (int)free_at.size(); free_at.push_back(0); }
    free_at[b] = vals[v].last_level;
    vals[v].buf = b;
  }
  if (free_at.size() > 1024) return ALOHA_OK;
  // ---- from here on the run is executed -------------------------------------------------------------------------------------
  *handled = true;
  vp->vl_bits = vl; vp->q = q; vp->mu = mu;
  vp->stats.instructions += ninstr;
  if (!nlive) return ALOHA_OK;
  if (free_at.size() > vp->flow_bufs) {
    HIPCK(vp, hipStreamSynchronize(vp->stream));
    if (vp->d_flow) { (void)hipFree(vp->d_flow); vp->d_flow = nullptr; vp->flow_bufs = 0; vp->epoch++; }   // recorded runs point into it
    const size_t want = std::max<size_t>(free_at.size(), 16);
    HIPCK(vp, hipMalloc(&vp->d_flow, (want * 8) << vp->max_logn));
    vp->flow_bufs = want;
  }
  auto addr = [&](int v) -> u64 * {
    const Val &x = vals[v];
    return x.fixed ? const_cast<u64 *>(x.fixed) : vp->d_flow + ((size_t)x.buf << vp->max_logn);
  };
  hipStream_t st = vp->stream;
  for (int lv = 1; lv <= max_level; lv++) {
    std::vector<LimbJob> fwd, inv;
    std::vector<aloha::k::EwJob> ew;
    std::vector<aloha::k::AutJob> aut;
    std::vector<aloha::k::DotArgs> sums;
    for (size_t i = 0; i < ops.size(); i++) {
      const Op &o = ops[i];
      if (!o.live || o.level != lv) continue;
      switch (o.kind) {
        case K_SUM: {
          aloha::k::DotArgs g{};
          int t = 0;
          for (const auto &x : o.vv) { g.a[t] = addr(x.first); g.b[t] = addr(x.second); t++; }
          for (const auto &x : o.lin) { g.a[t] = addr(x.first); g.b[t] = nullptr; g.s[t] = x.second; g.sp[t] = aloha::nt::shoup(x.second, o.q); t++; }
          if (o.cst) { g.a[t] = nullptr; g.b[t] = nullptr; g.s[t] = o.cst; t++; }
          g.nterms = t; g.init = nullptr; g.d = addr(o.dst); g.m = modctx(o.q);
          sums.push_back(g);
          break;
        }
        case K_NTT: case K_INTT: {
          uint32_t slot;
          aloha_status s = get_slot(vp, o.q, logn, &slot);
          if (s) return s;
          LimbJob jb{addr(o.a), addr(o.dst), slot, 0, {0, 0, 0, 0}, 0};
          if (o.kind == K_INTT && o.scaled) {
            const auto &tt = vp->slots[slot].dev.tw;
            jb.scaled = 1;
            jb.sc[0] = aloha::nt::mulmod(tt.w1n_inv, o.mulc, o.q); jb.sc[1] = aloha::nt::shoup(jb.sc[0], o.q);
            jb.sc[2] = aloha::nt::mulmod(tt.n_inv, o.mulc, o.q);   jb.sc[3] = aloha::nt::shoup(jb.sc[2], o.q);
            jb.addc = o.addc;
          }
          (o.kind == K_NTT ? fwd : inv).push_back(jb);
          break;
        }
        case K_EW: {
          aloha::k::EwJob jb{};
          jb.d = addr(o.dst); jb.a = addr(o.a); jb.b = o.b >= 0 ? addr(o.b) : jb.a;
          jb.s = o.scalar; jb.s_shoup = aloha::nt::shoup(o.scalar, o.q); jb.m = modctx(o.q); jb.op = o.ewop;
          ew.push_back(jb);
          break;
        }
        case K_AUT: {
          aloha::k::AutJob jb{};
          jb.out = addr(o.dst); jb.in = addr(o.a); jb.q = o.q; jb.k = o.k; jb.k_inv = o.kinv; jb.roll = o.roll;
          aut.push_back(jb);
          break;
        }
        case K_COPY: {
          u64 *cd = addr(o.dst); const u64 *cs = addr(o.a); const size_t cb = (size_t)n * 8;
          if (cd != cs) HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(cd, cs, cb, hipMemcpyDeviceToDevice, st); }));
          break;
        }
      }
    }
    // limbs that share a prime adjacent: one twiddle block per bundle of the transform kernels
    auto by_prime = [](const LimbJob &x, const LimbJob &y) { return x.prime < y.prime; };
    std::stable_sort(fwd.begin(), fwd.end(), by_prime);
    std::stable_sort(inv.begin(), inv.end(), by_prime);
    if (!fwd.empty()) { aloha_status s = run_transforms(vp, fwd, logn, false); if (s) return s; }
    if (!inv.empty()) { aloha_status s = run_transforms(vp, inv, logn, true); if (s) return s; }
    if (!ew.empty()) {
      const void *dj = nullptr;
      aloha_status s = device_blob(vp, ew.data(), ew.size() * sizeof(ew[0]), &dj);
      if (s) return s;
      const uint32_t nj = (uint32_t)ew.size(), nn = n;
      HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::ew_batch(st, (const aloha::k::EwJob *)dj, nj, nn); }));
    }
    if (!sums.empty()) {
      const void *dj = nullptr;
      aloha_status s = device_blob(vp, sums.data(), sums.size() * sizeof(sums[0]), &dj);
      if (s) return s;
      const uint32_t nj = (uint32_t)sums.size(), nn = n;
      HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::dot_batch(st, (const aloha::k::DotArgs *)dj, nj, nn); }));
    }
    if (!aut.empty()) {
      const void *dj = nullptr;
      aloha_status s = device_blob(vp, aut.data(), aut.size() * sizeof(aut[0]), &dj);
      if (s) return s;
      const uint32_t nj = (uint32_t)aut.size();
      const unsigned bl = logn;
      // (the single-pass gather: at these sizes a launch is its latency, and the two coalesced passes are two launches)
      HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::aut_batch(st, (const aloha::k::AutJob *)dj, nj, bl, false); }));
    }
  }
  vp->stats.flow_runs++;
  return ALOHA_OK;
}

}  // namespace

extern "C" aloha_status aloha_vp_run(aloha_vp *vp, uint32_t pc, uint64_t src0_row, uint64_t src1_row,
                                     uint64_t rslt_row, uint64_t ksk_row, uint64_t step) {
  if (!vp) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  const uint64_t csr[4] = {src0_row, src1_row, rslt_row, ksk_row};
  hipStream_t st = vp->stream;
  const bool fuse = !(vp->flags & ALOHA_FLAG_NO_FUSION);
  // ---- replay of a recorded run (see aloha_vp::RunPlan) ---------------------------------------------
  struct RecGuard { aloha_vp *v; ~RecGuard() { v->rec = nullptr; } } rec_guard{vp};
  vp->rec = nullptr;
  aloha_vp::RunPlan fresh;
  uint64_t run_hash = 0;
  const uint64_t instructions0 = vp->stats.instructions;
  if (fuse && vp->run_cache_on && pc < vp->iram_depth) {
    uint32_t end = pc;
    while (end < vp->iram_depth && ((vp->inst[end] >> 26) & 0x3f) != 0x10) end++;      // break  expander.v:141-151
    if (end < vp->iram_depth) {
      const uint32_t len = end - pc + 1;
      uint64_t h = 0xcbf29ce484222325ull;
      auto mix = [&](uint64_t x) { h = (h ^ x) * 0x100000001b3ull; };
      for (uint32_t i = pc; i <= end; i++) { mix(vp->inst[i]); mix(vp->imm[i]); }
      for (int i = 0; i < 4; i++) mix(csr[i]);
      mix(step); mix(vp->vl_bits); mix(vp->q); mix(vp->mu); mix(pc); mix(vp->scratch_row); mix(vp->scratch_rows);
      run_hash = h;
      auto range = vp->run_cache.equal_range(h);
      for (auto it = range.first; it != range.second; ++it) {
        aloha_vp::RunPlan &pl = it->second;
        if (pl.epoch != vp->epoch || pl.pc != pc || pl.inst.size() != len || pl.step != step || pl.vl_in != vp->vl_bits ||
            pl.q_in != vp->q || pl.mu_in != vp->mu || memcmp(pl.csr, csr, sizeof csr) != 0 ||
            pl.scratch_row != vp->scratch_row || pl.scratch_rows != vp->scratch_rows ||
            memcmp(pl.inst.data(), &vp->inst[pc], len * sizeof(uint32_t)) != 0 ||
            memcmp(pl.imm.data(), &vp->imm[pc], len * sizeof(uint64_t)) != 0)
          continue;
#ifdef ALOHA_EXPERIMENTS
        if (vp->graphs_on && !pl.graph && !pl.graph_failed && ++pl.replays >= 2 && !vp->fused_stats) {
          // capture the list once; a failure (an op that cannot be captured) leaves it a list for good
          const uint64_t l0 = vp->stats.kernel_launches, f0 = vp->stats.fused_launches;
          bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
          if (ok) {
            vp->capturing = true;
            for (const auto &op : pl.ops) ok = (op() == hipSuccess) && ok;
            vp->capturing = false;
            hipGraph_t g = nullptr;
            ok = (hipStreamEndCapture(st, &g) == hipSuccess) && ok && g;
            if (ok) ok = hipGraphInstantiate(&pl.graph, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
          }
          pl.launches = vp->stats.kernel_launches - l0;
          vp->stats.kernel_launches = l0;                // nothing ran during the capture
          vp->stats.fused_launches = f0;
          if (!ok) { pl.graph = nullptr; pl.graph_failed = true; (void)hipGetLastError(); }
        }
        if (pl.graph) {
          const hipError_t e = hipGraphLaunch(pl.graph, st);
          if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "graph launch of a recorded run failed: %s", hipGetErrorString(e));
          vp->stats.kernel_launches += pl.launches;
          vp->stats.graph_launches++;
        } else
#endif
        {
          for (const auto &op : pl.ops) {
            const hipError_t e = op();
            if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "replayed launch failed: %s", hipGetErrorString(e));
          }
        }
        vp->vl_bits = pl.vl_out; vp->q = pl.q_out; vp->mu = pl.mu_out;
        vp->stats.instructions += pl.instructions;
        return ALOHA_OK;
      }
      fresh.inst.assign(vp->inst.begin() + pc, vp->inst.begin() + end + 1);
      fresh.imm.assign(vp->imm.begin() + pc, vp->imm.begin() + end + 1);
      memcpy(fresh.csr, csr, sizeof csr);
      fresh.step = step; fresh.vl_in = vp->vl_bits; fresh.q_in = vp->q; fresh.mu_in = vp->mu; fresh.pc = pc;
      fresh.scratch_row = vp->scratch_row; fresh.scratch_rows = vp->scratch_rows;
      vp->rec = &fresh;
    }
  }
  auto finish_recording = [&]() {
    if (!vp->rec) return;
    vp->rec = nullptr;
    fresh.epoch = vp->epoch;
    fresh.vl_out = vp->vl_bits; fresh.q_out = vp->q; fresh.mu_out = vp->mu;
    fresh.instructions = vp->stats.instructions - instructions0;
    if (vp->run_cache.size() >= 1024) { vp->run_cache.clear(); }
    vp->run_cache.emplace(run_hash, std::move(fresh));
  };
  {
    // (q, N) pairs this segment may set that have no twiddle set yet: if they would overflow the cache, start a new
    // generation NOW -- before any job of this run holds a slot index
    std::set<std::pair<uint64_t, uint64_t>> fresh_sets;
    uint64_t q_scan = vp->q, vl_scan = vp->vl_bits;
    auto note = [&]() {
      unsigned ln = 0;
      if (q_scan && vl_logn(vl_scan, vp->max_logn, &ln) == ALOHA_OK && !vp->slot_of.count(std::make_pair(q_scan, ln)))
        fresh_sets.insert(std::make_pair(q_scan, (uint64_t)ln));
    };
    note();                                                   // the pair inherited from the previous run: a program may transform before its first vsetq
    for (uint32_t i = pc; i < vp->iram_depth; i++) {
      const Decoded d = decode_at(vp, i);                     // (the decoder of the sequencer below: funct6 selects, expander.v:123-130)
      if (d.funct6 == 0x10) break;
      if (d.funct6 == 0x08) q_scan = d.imm;
      else if (d.funct6 == 0x04) vl_scan = d.imm;
      else continue;
      note();
    }
    if (vp->slots.size() + fresh_sets.size() > vp->max_slots) {
      aloha_status s = make_room_for_twiddles(vp, std::min<size_t>(fresh_sets.size(), vp->max_slots));
      if (s) return s;
    }
  }
  const uint64_t epoch0 = vp->epoch;
  if (fuse && vp->flow_on && !(vp->flags & ALOHA_FLAG_RADIX2_ONLY)) {   // small rings: by the program's dependency graph (run_flow above)
    bool handled = false;
    const aloha_status sf = run_flow(vp, pc, csr, step, &handled);
    if (handled) {
      if (!sf && vp->epoch == epoch0) finish_recording();
      return sf;
    }
  }
  const u64 *alias[ALOHA_NVREG] = {nullptr};                // non-null: the register's value is memory at this address
  // Pending batch: deferred memory-to-memory jobs of ONE kind, launched together by flush().
  enum { PK_NONE, PK_NTT, PK_AUT, PK_EW, PK_DOT, PK_PROD } pend_kind = PK_NONE;
  std::vector<LimbJob> batch;                               // PK_NTT
  std::vector<aloha::k::ProdJob> prod_batch;                // PK_PROD: whole products iNTT(NTT(a) .* NTT(b)) at N = 2^14
  std::vector<aloha::k::AutJob> aut_batch;                  // PK_AUT
  std::vector<aloha::k::DotArgs> dot_batch;                 // PK_DOT: lazy dot products stored straight to memory
  const u64 *dot_w_lo = nullptr, *dot_w_hi = nullptr, *dot_a_lo = nullptr, *dot_a_hi = nullptr, *dot_b_lo = nullptr,
            *dot_b_hi = nullptr;                           // boxes of the pending sums' destinations / first / second operands
  std::vector<aloha::k::EwJob> ew_batch;                    // PK_EW
  std::vector<std::pair<const u64 *, const u64 *>> pend_rw; // (read range, written range) of every pending job
  // bounding boxes of the pending reads / writes: a new job that misses both boxes conflicts with
  // nothing, so loops over disjoint regions (the usual shape) join in O(1) instead of O(pending)
  const u64 *box_r_lo = nullptr, *box_r_hi = nullptr, *box_w_lo = nullptr, *box_w_hi = nullptr;
  auto note_rw = [&](const u64 *r, const u64 *w, size_t n) {
    pend_rw.emplace_back(r, w);
    if (r) { if (!box_r_lo || r < box_r_lo) box_r_lo = r; if (!box_r_hi || r + n > box_r_hi) box_r_hi = r + n; }
    if (!box_w_lo || w < box_w_lo) box_w_lo = w;
    if (!box_w_hi || w + n > box_w_hi) box_w_hi = w + n;
  };
  auto hits = [](const u64 *p, size_t n, const u64 *lo, const u64 *hi) { return p && lo && p < hi && lo < p + n; };
  bool batch_inv = false;
  unsigned batch_logn = 0;
  // Lazy dot products: register r stands for sum_t a_t .* b_t (mod lazy_q[r]) over memory
  // operands -- the key-switch inner product (keyswitch.mem:46-79) evaluated in one pass.
  // (second == nullptr: the term is a_t times the scalar s, from vfqmul.vs; first == nullptr too: the constant s)
  struct LazyTerm { const u64 *first, *second; u64 s; };
  std::vector<LazyTerm> lazy[ALOHA_NVREG];
  uint64_t lazy_q[ALOHA_NVREG] = {0};

  auto rd = [&](unsigned r) -> const u64 * { return alias[r] ? alias[r] : vreg(vp, r); };
  // Sums stored to memory, launched together.  Jobs that are sums of scaled operands over the SAME operands (a basis
  // conversion: one digit under every target modulus) leave as groups -- the operands are read once per element instead of
  // once per output.
  auto launch_dots = [&](const std::vector<aloha::k::DotArgs> &jobs, unsigned lg) -> aloha_status {
    aloha_status s = ALOHA_OK;
    if (jobs.empty()) return s;
    std::vector<aloha::k::SumGroup> groups;
    std::vector<aloha::k::SumJob> sjobs;
    std::vector<aloha::k::DotArgs> rest;
    if (fuse && jobs.size() >= 4) {
      std::map<std::array<const u64 *, aloha::k::kSumTerms>, std::vector<size_t>> by_ops;
      std::vector<char> grouped(jobs.size(), 0);
      for (size_t i = 0; i < jobs.size(); i++) {
        const auto &g = jobs[i];
        std::array<const u64 *, aloha::k::kSumTerms> key{};
        int nv = 0;
        bool ok = !g.init;
        for (int t = 0; ok && t < g.nterms; t++) {
          if (g.b[t]) ok = false;                                 // a vector x vector term
          else if (g.a[t]) { if (nv == aloha::k::kSumTerms) ok = false; else key[nv++] = g.a[t]; }
        }
        if (ok && nv) by_ops[key].push_back(i);
      }
      for (const auto &kv : by_ops) {
        if (kv.second.size() < 4) continue;
        // a group's outputs are a serial loop inside one thread: at most kPerGroup of them per group, so that
        // there are enough workgroups in flight (the operands are then re-read once per sub-group, from L2)
        constexpr size_t kPerGroup = 4;
        for (size_t j0 = 0; j0 < kv.second.size(); j0 += kPerGroup) {
          aloha::k::SumGroup sg{};
          int nv = 0;
          while (nv < aloha::k::kSumTerms && kv.first[nv]) { sg.a[nv] = kv.first[nv]; nv++; }
          sg.nterms = nv; sg.first = (uint32_t)sjobs.size();
          sg.count = (uint32_t)std::min(kPerGroup, kv.second.size() - j0);
          for (size_t jj = j0; jj < j0 + sg.count; jj++) {
            const size_t i = kv.second[jj];
            const auto &g = jobs[i];
            aloha::k::SumJob sj{};
            int v = 0;
            for (int t = 0; t < g.nterms; t++) {
              if (g.a[t]) { sj.s[v] = g.s[t]; sj.sp[v] = g.sp[t]; v++; }
              else { sj.cst += g.s[t]; if (sj.cst >= g.m.q) sj.cst -= g.m.q; }
            }
            sj.d = g.d; sj.m = g.m;
            sjobs.push_back(sj);
            grouped[i] = 1;
          }
          groups.push_back(sg);
        }
      }
      if (!groups.empty())
        for (size_t i = 0; i < jobs.size(); i++) if (!grouped[i]) rest.push_back(jobs[i]);
    }
    const std::vector<aloha::k::DotArgs> &plain = groups.empty() ? jobs : rest;
    if (!groups.empty()) {
      // one blob: groups first, then jobs (device_blob hands out one region per call)
      std::vector<unsigned char> blob(groups.size() * sizeof(groups[0]) + sjobs.size() * sizeof(sjobs[0]));
      memcpy(blob.data(), groups.data(), groups.size() * sizeof(groups[0]));
      memcpy(blob.data() + groups.size() * sizeof(groups[0]), sjobs.data(), sjobs.size() * sizeof(sjobs[0]));
      const void *dg = nullptr;
      s = device_blob(vp, blob.data(), blob.size(), &dg);
      if (!s) {
        const uint32_t ng = (uint32_t)groups.size(), nn = 1u << lg;
        const aloha::k::SumJob *djobs = (const aloha::k::SumJob *)((const unsigned char *)dg + groups.size() * sizeof(groups[0]));
        hipError_t e = issue(vp, [=]() -> hipError_t {
          vp->stats.kernel_launches++;
          return aloha::k::sum_groups(st, (const aloha::k::SumGroup *)dg, ng, djobs, nn);
        });
        if (e != hipSuccess) s = fail(vp, ALOHA_ERR_DEVICE, "sum_groups: %s", hipGetErrorString(e));
      }
    }
    if (!s && !plain.empty()) {
      const void *dj = nullptr;
      s = device_blob(vp, plain.data(), plain.size() * sizeof(plain[0]), &dj);
      if (!s) {
        const uint32_t nj = (uint32_t)plain.size(), nn = 1u << lg;
        hipError_t e = issue(vp, [=]() -> hipError_t {
          vp->stats.kernel_launches++;
          return aloha::k::dot_batch(st, (const aloha::k::DotArgs *)dj, nj, nn);
        });
        if (e != hipSuccess) s = fail(vp, ALOHA_ERR_DEVICE, "dot_batch: %s", hipGetErrorString(e));
      }
    }
    return s;
  };
  // ---- staged batches (the fused key-switch phases at N = 2^16) ------------------------------------------------------------
  // A phase program is a chain  sums -> forward transforms -> sums  (keyswitch.mem:9-14 -> :10,:14 -> :46-79, and
  // :86-91 -> :94-99 -> :102-117).  Instead of launching a batch when a job of another kind arrives, a batch of sums that
  // is followed by forward transforms is HELD (stage 0), and so is the batch of transforms when sums follow it (stage 1):
  // stages keep program order, so holding changes nothing by itself.  At the flush the stages are matched:
  //   * a sum whose store is overwritten by the in-place transform of the same rows becomes the column pass's input
  //     (SumSrc: never materialised -- exact, the store was dead);
  //   * inner products  sum_d X_d .* KEY_{d,c}  over transformed limbs X_d become the row pass's output (RowDotGroup), and
  //     a sum of scaled operands with ONE transformed limb among them the row pass's epilogue (RowEpi).  The transformed
  //     limbs themselves are still stored, unless they lie in the caller's scratch window (aloha_vp_set_scratch) and no
  //     later instruction of the run loads them.
  // Anything that does not match is launched as before, in stage order.
  std::vector<aloha::k::DotArgs> held_sums;
  std::vector<LimbJob> held_ntt;
  auto in_scratch = [&](const u64 *p_, size_t n_) {
    const u64 *lo = vp->d_spm + vp->scratch_row * kLanes, *hi = lo + vp->scratch_rows * kLanes;
    return vp->scratch_rows && p_ >= lo && p_ + n_ <= hi;
  };
  // are rows of [p_, p_ + n_) still to be read -- through a register that aliases them, a lazy sum over them, or a load by
  // the instruction being sequenced (pc) or a later one?
  auto loaded_later = [&](const u64 *p_, uint32_t n_) {
    for (unsigned r = 0; r < ALOHA_NVREG; r++) {
      bool refers = alias[r] && overlap(alias[r], p_, n_);
      for (const auto &t : lazy[r]) refers = refers || (t.first && overlap(t.first, p_, n_)) || (t.second && overlap(t.second, p_, n_));
      if (refers && !dead_from(vp, pc, r)) return true;
    }
    for (uint32_t i = pc; i < vp->iram_depth; i++) {
      const Decoded d = decode_at(vp, i);
      if (d.funct6 == 0x10) return false;
      if (d.funct6 == 0x04) return true;                      // the vector length changes: not followed further
      if (d.funct6 != 0x03) continue;
      const u64 *r = mem_rows(vp, d.imm, true, csr, n_ / kLanes);
      if (r && overlap(r, p_, n_)) return true;
    }
    return true;
  };
  auto launch_staged = [&]() -> aloha_status {
    std::vector<aloha::k::DotArgs> S, D;
    std::vector<LimbJob> T;
    S.swap(held_sums);
    if (!held_ntt.empty()) {
      T.swap(held_ntt);
      if (pend_kind == PK_DOT) D.swap(dot_batch);
    } else if (pend_kind == PK_NTT && !batch_inv) {
      T.swap(batch);
    }
    const unsigned lg = batch_logn;
    const uint32_t nn = 1u << lg;
    const size_t nT = T.size();
    aloha_status s = ALOHA_OK;
    // Fusing moves work in time: a fused sum reads its operands while the transforms run, a fused inner product or
    // epilogue reads its operands and writes its result while the transforms run.  Besides the one intended dependency
    // nothing of the transforms' batch may touch those rows.
    auto t_writes = [&](const u64 *p_, const u64 *except) {
      for (const LimbJob &j : T) if (j.dst != except && overlap(p_, j.dst, nn)) return true;
      return false;
    };
    auto t_touches = [&](const u64 *p_) {
      for (const LimbJob &j : T) if (overlap(p_, j.dst, nn) || overlap(p_, j.src, nn)) return true;
      return false;
    };
    // -- stage 0 -> 1: sums that only feed an in-place transform ------------------------------------------------------------
    std::vector<int> pro_of(nT, -1);
    std::vector<char> s_used(S.size(), 0);
    bool any = false;
    if (nT && lg == 16 && !S.empty()) {
      std::unordered_map<const u64 *, size_t> by_dst;
      for (size_t i = 0; i < S.size(); i++) by_dst[S[i].d] = i;
      for (size_t t = 0; t < nT; t++) {
        if (T[t].src != T[t].dst) continue;
        auto it = by_dst.find(T[t].src);
        if (it == by_dst.end()) continue;
        const auto &g = S[it->second];
        int nv = 0;
        bool ok = !g.init && g.m.q == vp->slots[T[t].prime].q;
        for (int k = 0; ok && k < g.nterms; k++) {
          if (g.b[k]) ok = false;
          else if (g.a[k]) ok = ++nv <= aloha::k::kSumTerms && !t_writes(g.a[k], nullptr);
        }
        if (!ok || !nv) continue;
        pro_of[t] = (int)it->second; s_used[it->second] = 1; any = true;
      }
    }
    // -- stage 1 -> 2: sums over transformed limbs -------------------------------------------------------------------------
    std::vector<int> t_use(nT, 0), epi_of(nT, -1);             // t_use: 1 = digit of an inner product, 2 = carries an epilogue
    std::vector<char> d_fused(D.size(), 0);
    struct Cand { std::vector<const u64 *> x, key; };
    std::vector<Cand> cand(D.size());
    std::map<std::vector<const u64 *>, std::vector<size_t>> dgroups;
    if (nT && lg == 16 && !D.empty()) {
      std::unordered_map<const u64 *, size_t> t_by_dst;
      for (size_t t = 0; t < nT; t++) t_by_dst[T[t].dst] = t;
      bool ok = true;
      for (size_t i = 0; i < D.size() && ok; i++) {
        const auto &g = D[i];
        int n_t = 0, nvv = 0, nsc = 0;
        for (int k = 0; k < g.nterms; k++) {
          n_t += (g.a[k] && t_by_dst.count(g.a[k])) + (g.b[k] && t_by_dst.count(g.b[k]));
          if (g.a[k] && g.b[k]) nvv++; else if (g.a[k]) nsc++;
        }
        if (!n_t) continue;                                      // does not read a transformed limb: an ordinary sum behind the transforms
        if (g.init || t_touches(g.d)) { ok = false; break; }
        for (int k = 0; k < g.nterms && ok; k++)                 // operands that merely overlap a transform's output
          for (const u64 *o : {g.a[k], g.b[k]}) if (o && !t_by_dst.count(o) && t_writes(o, nullptr)) ok = false;
        if (!ok) break;
        if (nvv == g.nterms && g.nterms <= aloha::k::kRowDotMax) {
          Cand &c = cand[i];
          for (int k = 0; k < g.nterms && ok; k++) {
            const bool at = t_by_dst.count(g.a[k]) != 0, bt = t_by_dst.count(g.b[k]) != 0;
            if (at && bt) ok = false;
            else if (bt) { c.x.push_back(g.b[k]); c.key.push_back(g.a[k]); }
            else { c.x.push_back(g.a[k]); c.key.push_back(g.b[k]); }
          }
          if (ok) dgroups[c.x].push_back(i);
        } else if (!nvv && n_t == 1 && nsc <= aloha::k::kEpiOps + 1) {
          size_t tt = 0;
          for (int k = 0; k < g.nterms; k++) if (g.a[k] && t_by_dst.count(g.a[k])) tt = t_by_dst[g.a[k]];
          if (t_use[tt] || g.m.q != vp->slots[T[tt].prime].q) ok = false;
          else { t_use[tt] = 2; epi_of[tt] = (int)i; d_fused[i] = 1; }
        } else ok = false;
      }
      for (auto &kv : dgroups) {
        if (!ok) break;
        if (kv.second.size() != 2 || D[kv.second[0]].m.q != D[kv.second[1]].m.q) { ok = false; break; }
        for (const u64 *x : kv.first) {
          auto it = t_by_dst.find(x);
          if (it == t_by_dst.end()) continue;
          if (t_use[it->second] || vp->slots[T[it->second].prime].q != D[kv.second[0]].m.q) { ok = false; break; }
          t_use[it->second] = 1;
        }
        if (ok) for (size_t i : kv.second) d_fused[i] = 1;
      }
      if (!ok) {                                                 // something reads a transformed limb in a way the kernels do not cover
        std::fill(t_use.begin(), t_use.end(), 0); std::fill(epi_of.begin(), epi_of.end(), -1);
        std::fill(d_fused.begin(), d_fused.end(), 0); dgroups.clear();
      } else if (!dgroups.empty() || std::count(t_use.begin(), t_use.end(), 2)) any = true;
    }
    if (!any) {                                                  // nothing to fuse: the stages as separate batches
      s = launch_dots(S, lg);
      if (!s && nT) s = run_transforms(vp, T, lg, false);
      if (!s) s = launch_dots(D, lg);
      return s;
    }
    {
      std::vector<aloha::k::DotArgs> rest;
      for (size_t i = 0; i < S.size(); i++) if (!s_used[i]) rest.push_back(S[i]);
      s = launch_dots(rest, lg);
      if (s) return s;
    }
    // limbs of inner products first (column pass only), the others behind them (column pass + row tiles)
    std::vector<size_t> order;
    for (size_t t = 0; t < nT; t++) if (t_use[t] == 1) order.push_back(t);
    const uint32_t n_dot = (uint32_t)order.size();
    if (n_dot >= 16) {
      // Their order is free (their row pass walks descriptors, not a tile list).  Limbs whose sums share operands -- one
      // digit lifted to many moduli -- are dealt out so that one XCD works through them back to back: the column
      // kernel maps limb 8s + x to XCD x (a placement observed, not promised: it only matters for speed), and the
      // operands of <= 2 digits (2 MiB each at N = 2^16, alpha = 4) then stay in that XCD's 4 MiB L2 instead of being
      // fetched again for every limb.
      auto key_of = [&](size_t t) {
        std::array<const u64 *, aloha::k::kSumTerms> key{};
        if (pro_of[t] >= 0) {
          const auto &g = S[pro_of[t]];
          int nv = 0;
          for (int k = 0; k < g.nterms; k++) if (g.a[k]) key[nv++] = g.a[k];
        }
        return key;
      };
      std::vector<size_t> rd(order);
      std::stable_sort(rd.begin(), rd.end(), [&](size_t a, size_t b) { return key_of(a) < key_of(b); });
      const size_t full = n_dot / 8 * 8, per = full / 8;
      for (size_t p = 0; p < full; p++) order[(p % per) * 8 + p / per] = rd[p];
      for (size_t p = full; p < n_dot; p++) order[p] = rd[p];
    }
    for (size_t t = 0; t < nT; t++) if (t_use[t] != 1) order.push_back(t);
    std::vector<size_t> pos_of(nT);
    for (size_t i = 0; i < nT; i++) pos_of[order[i]] = i;
    std::vector<LimbJob> J(2 * nT);
    std::vector<aloha::k::SumSrc> sums;
    std::vector<aloha::k::RowEpi> epis;
    for (size_t i = 0; i < nT; i++) {
      const size_t t = order[i];
      LimbJob jb = T[t];
      if (pro_of[t] >= 0) {
        const auto &g = S[pro_of[t]];
        aloha::k::SumSrc ss{};
        for (int k = 0; k < g.nterms; k++) {
          if (g.a[k]) { ss.a[ss.nterms] = g.a[k]; ss.s[ss.nterms] = g.s[k]; ss.sp[ss.nterms] = g.sp[k]; ss.nterms++; }
          else { ss.cst += g.s[k]; if (ss.cst >= g.m.q) ss.cst -= g.m.q; }
        }
        sums.push_back(ss);
        jb.pro = (uint32_t)sums.size();
      }
      if (epi_of[t] >= 0) {
        const auto &g = D[epi_of[t]];
        aloha::k::RowEpi e{};
        for (int k = 0; k < g.nterms; k++) {
          if (g.a[k] == jb.dst) { e.s[0] = g.s[k]; e.sp[0] = g.sp[k]; }
          else if (g.a[k]) { e.a[e.nops] = g.a[k]; e.s[e.nops + 1] = g.s[k]; e.sp[e.nops + 1] = g.sp[k]; e.nops++; }
          else { e.cst += g.s[k]; if (e.cst >= g.m.q) e.cst -= g.m.q; }
        }
        e.out = g.d;
        e.store_t = (in_scratch(jb.dst, nn) && !loaded_later(jb.dst, nn)) ? 0 : 1;
        epis.push_back(e);
        jb.epi = (uint32_t)epis.size();
      }
      J[i] = jb;
      J[nT + i] = jb; J[nT + i].src = jb.dst;
    }
    std::vector<aloha::k::RowDotGroup> rgroups;
    for (const auto &kv : dgroups) {
      const Cand &c0 = cand[kv.second[0]], &c1 = cand[kv.second[1]];
      aloha::k::RowDotGroup rg{};
      std::unordered_map<const u64 *, size_t> t_by_dst;
      for (size_t t = 0; t < nT; t++) if (t_use[t] == 1) t_by_dst[T[t].dst] = t;
      for (int pass = 0; pass < 2; pass++)                        // transformed digits first, then the ones in NTT form already
        for (size_t k = 0; k < c0.x.size(); k++) {
          const bool is_t = t_by_dst.count(c0.x[k]) != 0;
          if (is_t != (pass == 0)) continue;
          const uint32_t d = rg.nt + rg.np;
          rg.x[d] = c0.x[k]; rg.key[d][0] = c0.key[k]; rg.key[d][1] = c1.key[k];
          if (is_t) {
            rg.prime = T[t_by_dst[c0.x[k]]].prime;
            rg.xout[d] = (in_scratch(c0.x[k], nn) && !loaded_later(c0.x[k], nn)) ? nullptr : const_cast<u64 *>(c0.x[k]);
            rg.nt++;
          } else rg.np++;
        }
      rg.out[0] = D[kv.second[0]].d; rg.out[1] = D[kv.second[1]].d;
      rgroups.push_back(rg);
    }
    std::vector<TileDesc> uni, glb;
    if (nT > n_dot) aloha::k::build_row_tiles(J.data() + nT, n_dot, (uint32_t)nT - n_dot, lg - 8, uni, glb);
    std::vector<TileDesc> tiles(uni);
    tiles.insert(tiles.end(), glb.begin(), glb.end());
    const void *dJ = nullptr, *dS = nullptr, *dE = nullptr, *dG = nullptr, *dT = nullptr;
    s = device_blob(vp, J.data(), J.size() * sizeof(J[0]), &dJ);
    if (!s && !sums.empty()) s = device_blob(vp, sums.data(), sums.size() * sizeof(sums[0]), &dS);
    if (!s && !epis.empty()) s = device_blob(vp, epis.data(), epis.size() * sizeof(epis[0]), &dE);
    if (!s && !rgroups.empty()) s = device_blob(vp, rgroups.data(), rgroups.size() * sizeof(rgroups[0]), &dG);
    if (!s && !tiles.empty()) s = device_blob(vp, tiles.data(), tiles.size() * sizeof(tiles[0]), &dT);
    if (s) return s;
    {
      const uint32_t cnt = (uint32_t)nT, n_uni = (uint32_t)uni.size(), n_glb = (uint32_t)glb.size(), n_grp = (uint32_t)rgroups.size();
      const uint32_t dot_rows = aloha::k::rowdot_rows(rgroups.data(), n_grp);
      const hipError_t e = issue(vp, [=]() -> hipError_t {
        uint64_t launches = 0;
        const LimbJob *dj = (const LimbJob *)dJ;
        const TileDesc *dt = (const TileDesc *)dT;
        hipError_t e = aloha::k::ntt_fwd16_cols(st, dj, vp->d_primes, cnt, (const aloha::k::SumSrc *)dS, &launches);
        if (e == hipSuccess) e = aloha::k::ntt_fwd16_rows(st, dj + cnt, vp->d_primes, dt, n_uni, dt + n_uni, n_glb, (const aloha::k::RowEpi *)dE, &launches);
        if (e == hipSuccess) e = aloha::k::ntt_fwd16_rowdot(st, (const aloha::k::RowDotGroup *)dG, n_grp, vp->d_primes, dot_rows, &launches);
        vp->stats.kernel_launches += launches;
        vp->stats.ks_fused_launches += launches;
        return e;
      });
      if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "fused key-switch transforms: %s", hipGetErrorString(e));
    }
    std::vector<aloha::k::DotArgs> rest;
    for (size_t i = 0; i < D.size(); i++) if (!d_fused[i]) rest.push_back(D[i]);
    return launch_dots(rest, lg);
  };
  // Element-wise products followed by sums: the tensor product of two ciphertexts is, per modulus, two `vfqmul.vv; vse`
  // (d0 = a0 b0, d2 = a1 b1) and a two-term sum (d1 = a0 b1 + a1 b0) over the same four limbs.  The batch of products is
  // held when sums follow; at the flush every (d0, d2, d1) triple over one quadruple of operands leaves as ONE job of
  // tensor_batch (each input read once: 7 limb passes instead of 11), the rest as before, in stage order.
  // (Round 4, measured and dropped: the tensor product formed INSIDE the inverse row pass that transforms its d2 -- HMult's first phase as
  // one launch less and d2 never read back: 32.6 us against 18.0 + 15.8 us for the two kernels.  The inverse transform of 24 limbs is a
  // latency-bound launch of 1.5 waves per SIMD; work moved into it is not hidden, it is serialised.  profiles/r04_tensor_in_inverse_row_pass.txt)
  std::vector<aloha::k::EwJob> held_ew;
  auto launch_tensor = [&]() -> aloha_status {
    std::vector<aloha::k::EwJob> E;
    std::vector<aloha::k::DotArgs> D;
    E.swap(held_ew);
    if (pend_kind == PK_DOT) D.swap(dot_batch);
    const unsigned lg = batch_logn;
    const uint32_t nn = 1u << lg;
    std::vector<aloha::k::TensorJob> T;
    std::vector<char> e_used(E.size(), 0), d_used(D.size(), 0);
    std::map<std::pair<const u64 *, const u64 *>, size_t> mul_of;   // (a, b) of a product -> its job
    for (size_t i = 0; i < E.size(); i++)
      if (E[i].op == aloha::k::EW_MUL_VV) mul_of[{E[i].a, E[i].b}] = i;
    auto e_writes = [&](const u64 *p_) { for (const auto &e : E) if (overlap(p_, e.d, nn)) return true; return false; };
    auto e_touches = [&](const u64 *p_) {
      for (const auto &e : E) if (overlap(p_, e.d, nn) || overlap(p_, e.a, nn) || overlap(p_, e.b, nn)) return true;
      return false;
    };
    for (size_t j = 0; j < D.size(); j++) {
      const auto &g = D[j];
      if (g.init || g.nterms != 2 || !g.a[0] || !g.b[0] || !g.a[1] || !g.b[1]) continue;
      // g = x0 .* y1 + x1 .* y0 with products (x0, y0) and (x1, y1) in the held batch (either operand order)
      size_t i0 = 0, i1 = 0;
      bool found = false;
      for (int sw = 0; sw < 4 && !found; sw++) {
        const u64 *x0 = (sw & 1) ? g.b[0] : g.a[0], *y1 = (sw & 1) ? g.a[0] : g.b[0];
        const u64 *x1 = (sw & 2) ? g.b[1] : g.a[1], *y0 = (sw & 2) ? g.a[1] : g.b[1];
        for (int ord = 0; ord < 4 && !found; ord++) {
          auto f0 = mul_of.find((ord & 1) ? std::make_pair(y0, x0) : std::make_pair(x0, y0));
          auto f1 = mul_of.find((ord & 2) ? std::make_pair(y1, x1) : std::make_pair(x1, y1));
          if (f0 == mul_of.end() || f1 == mul_of.end() || f0->second == f1->second || e_used[f0->second] || e_used[f1->second]) continue;
          const auto &e0 = E[f0->second], &e1 = E[f1->second];
          if (e0.m.q != g.m.q || e1.m.q != g.m.q) continue;
          // fusing reads the four operands while d0, d2 and d1 are written: none of the batch's outputs may be an operand
          // of the sum, and the sum's output may not touch anything the batch reads or writes
          if (e_writes(x0) || e_writes(x1) || e_writes(y0) || e_writes(y1) || e_touches(g.d)) continue;
          aloha::k::TensorJob t{};
          t.a0 = x0; t.a1 = x1; t.b0 = y0; t.b1 = y1; t.d0 = e0.d; t.d2 = e1.d; t.d1 = g.d; t.m = g.m;
          T.push_back(t);
          e_used[f0->second] = e_used[f1->second] = 1; d_used[j] = 1;
          i0 = f0->second; i1 = f1->second; found = true;
        }
      }
      (void)i0; (void)i1;
    }
    aloha_status s = ALOHA_OK;
    std::vector<aloha::k::EwJob> e_rest;
    for (size_t i = 0; i < E.size(); i++) if (!e_used[i]) e_rest.push_back(E[i]);
    if (!e_rest.empty()) {
      const void *dj = nullptr;
      s = device_blob(vp, e_rest.data(), e_rest.size() * sizeof(e_rest[0]), &dj);
      if (s) return s;
      const uint32_t nj = (uint32_t)e_rest.size();
      const hipError_t e = issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::ew_batch(st, (const aloha::k::EwJob *)dj, nj, nn); });
      if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "ew_batch: %s", hipGetErrorString(e));
    }
    if (!T.empty()) {
      const void *dj = nullptr;
      s = device_blob(vp, T.data(), T.size() * sizeof(T[0]), &dj);
      if (s) return s;
      const uint32_t nj = (uint32_t)T.size();
      const hipError_t e = issue(vp, [=]() -> hipError_t {
        vp->stats.kernel_launches++; vp->stats.ks_fused_launches++;
        return aloha::k::tensor_batch(st, (const aloha::k::TensorJob *)dj, nj, nn);
      });
      if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "tensor_batch: %s", hipGetErrorString(e));
    }
    std::vector<aloha::k::DotArgs> d_rest;
    for (size_t j = 0; j < D.size(); j++) if (!d_used[j]) d_rest.push_back(D[j]);
    return launch_dots(d_rest, lg);
  };
  auto reset_pending = [&]() {
    pend_rw.clear();
    box_r_lo = box_r_hi = box_w_lo = box_w_hi = nullptr;
    dot_w_lo = dot_w_hi = dot_a_lo = dot_a_hi = dot_b_lo = dot_b_hi = nullptr;
    pend_kind = PK_NONE;
  };
  const bool stage_ok = fuse && !(vp->flags & ALOHA_FLAG_RADIX2_ONLY) && !getenv("ALOHA_NO_STAGES");   // (the variable: A/B timing switch)
  auto flush = [&]() -> aloha_status {
    aloha_status s = ALOHA_OK;
    if (!held_ew.empty()) s = launch_tensor();                              // (takes the pending sums with it)
    else if (!held_sums.empty() || !held_ntt.empty()) s = launch_staged();  // (takes the pending batch with it when that is the next stage)
    if (s) {
    } else if (pend_kind == PK_NTT && !batch.empty()) {
      s = run_transforms(vp, batch, batch_logn, batch_inv);
    } else if (pend_kind == PK_AUT && !aut_batch.empty()) {
      const void *dj = nullptr;
      s = device_blob(vp, aut_batch.data(), aut_batch.size() * sizeof(aut_batch[0]), &dj);
      if (!s) {
        bool two_pass = !getenv("ALOHA_AUT_GATHER");                    // (the variable is an A/B timing switch)
        for (const auto &j : aut_batch) two_pass = two_pass && !j.roll;
        const uint32_t nj = (uint32_t)aut_batch.size();
        const unsigned bl = batch_logn;
        hipError_t e = issue(vp, [=]() -> hipError_t {
          vp->stats.kernel_launches += (two_pass && bl >= 12) ? 2 : 1;
          return aloha::k::aut_batch(st, (const aloha::k::AutJob *)dj, nj, bl, two_pass);
        });
        if (e != hipSuccess) s = fail(vp, ALOHA_ERR_DEVICE, "aut_batch: %s", hipGetErrorString(e));
      }
    } else if (pend_kind == PK_EW && !ew_batch.empty()) {
      const void *dj = nullptr;
      s = device_blob(vp, ew_batch.data(), ew_batch.size() * sizeof(ew_batch[0]), &dj);
      if (!s) {
        const uint32_t nj = (uint32_t)ew_batch.size(), nn = 1u << batch_logn;
        hipError_t e = issue(vp, [=]() -> hipError_t {
          vp->stats.kernel_launches++;
          return aloha::k::ew_batch(st, (const aloha::k::EwJob *)dj, nj, nn);
        });
        if (e != hipSuccess) s = fail(vp, ALOHA_ERR_DEVICE, "ew_batch: %s", hipGetErrorString(e));
      }
    }
    else if (pend_kind == PK_DOT && !dot_batch.empty()) {
      s = launch_dots(dot_batch, batch_logn);
    }
    else if (pend_kind == PK_PROD && !prod_batch.empty()) {
      if (!vp->d_prod_ctl) {                                   // first product of this handle: scratch (tag 0 everywhere) and the status word
        const size_t ww = aloha::k::poly_product_work_words() * 8 + (1u << 20);   // (+ 1 MiB: clock stamps of trace builds)
        const size_t fb = (size_t)2 * aloha::k::kProdMaxLimbs * 8 << 14;          // the five-launch form: two transformed operands per job
        if (hipMalloc(&vp->d_prod_work, ww) != hipSuccess || hipMalloc(&vp->d_prod_ctl, 64) != hipSuccess || hipMalloc(&vp->d_prod_fb, fb) != hipSuccess ||
            hipMemsetAsync(vp->d_prod_work, 0, ww, st) != hipSuccess || hipMemsetAsync(vp->d_prod_ctl, 0, 64, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
          if (vp->d_prod_work) (void)hipFree(vp->d_prod_work);
          if (vp->d_prod_ctl) (void)hipFree(vp->d_prod_ctl);
          if (vp->d_prod_fb) (void)hipFree(vp->d_prod_fb);
          vp->d_prod_work = nullptr; vp->d_prod_ctl = nullptr; vp->d_prod_fb = nullptr;
          s = fail(vp, ALOHA_ERR_DEVICE, "polynomial product: scratch allocation failed");
        }
        memset(vp->prod_tag, 0, sizeof vp->prod_tag);
      }
      for (size_t j0 = 0; !s && j0 < prod_batch.size(); j0 += aloha::k::kProdMaxLimbs) {
        const uint32_t nj = (uint32_t)std::min<size_t>(aloha::k::kProdMaxLimbs, prod_batch.size() - j0);
        const void *dj = nullptr;
        s = device_blob(vp, prod_batch.data() + j0, nj * sizeof(prod_batch[0]), &dj);
        if (s) break;
        // The same products as five ordinary launches -- both operands' forward transforms as one batch, the products, the inverse
        // transforms -- prepared beside the one-launch form: which of the two runs is decided when the launch is ISSUED (every
        // replay of a recorded run decides again): the one-launch form unless another handle of this device has such a launch in
        // flight or this handle's status word was ever set (prod_admit above).
        const uint32_t nn = 1u << batch_logn, cnt2 = 2 * nj;
        std::vector<LimbJob> F(2 * (size_t)cnt2), I(2 * (size_t)nj);
        std::vector<aloha::k::EwJob> E(nj);
        for (uint32_t i = 0; i < nj; i++) {
          const aloha::k::ProdJob &pj = prod_batch[j0 + i];
          u64 *wa = vp->d_prod_fb + ((size_t)(2 * i) << batch_logn), *wb = wa + nn;
          F[2 * i] = LimbJob{pj.a, wa, pj.prime, 0, {0, 0, 0, 0}, 0};
          F[2 * i + 1] = LimbJob{pj.b, wb, pj.prime, 0, {0, 0, 0, 0}, 0};
          I[i] = LimbJob{wa, pj.d, pj.prime, 0, {0, 0, 0, 0}, 0};
          aloha::k::EwJob e{};
          e.d = wa; e.a = wa; e.b = wb; e.m = vp->slots[pj.prime].dev.m; e.op = (int)aloha::k::EW_MUL_VV;
          E[i] = e;
        }
        for (uint32_t i = 0; i < cnt2; i++) { F[cnt2 + i] = F[i]; F[cnt2 + i].src = F[i].dst; }
        for (uint32_t i = 0; i < nj; i++) { I[nj + i] = I[i]; I[nj + i].src = I[i].dst; }
        std::vector<TileDesc> fu, fg, iu, ig;
        aloha::k::build_row_tiles(F.data() + cnt2, 0, cnt2, batch_logn - 8, fu, fg);
        aloha::k::build_row_tiles(I.data(), 0, nj, batch_logn - 8, iu, ig);
        std::vector<TileDesc> ft(fu), it(iu);
        ft.insert(ft.end(), fg.begin(), fg.end());
        it.insert(it.end(), ig.begin(), ig.end());
        const void *dF = nullptr, *dI = nullptr, *dE = nullptr, *dFt = nullptr, *dIt = nullptr;
        s = device_blob(vp, F.data(), F.size() * sizeof(F[0]), &dF);
        if (!s) s = device_blob(vp, I.data(), I.size() * sizeof(I[0]), &dI);
        if (!s) s = device_blob(vp, E.data(), E.size() * sizeof(E[0]), &dE);
        if (!s && !ft.empty()) s = device_blob(vp, ft.data(), ft.size() * sizeof(ft[0]), &dFt);
        if (!s && !it.empty()) s = device_blob(vp, it.data(), it.size() * sizeof(it[0]), &dIt);
        if (s) break;
        const uint32_t nfu = (uint32_t)fu.size(), nfg = (uint32_t)fg.size(), niu = (uint32_t)iu.size(), nig = (uint32_t)ig.size();
        const unsigned lg = batch_logn;
        const unsigned single_max = (vp->flags & ALOHA_FLAG_NTT_TWO_PASS) ? 0u : (vp->flags & ALOHA_FLAG_NTT_SINGLE_14) ? 14u : 13u;
        const hipError_t e = issue(vp, [=]() -> hipError_t {
          if (vp->capturing) return hipErrorStreamCaptureUnsupported;   // (the tag and the admission are decided per launch: not for a captured graph)
          uint64_t launches = 0;
          if (vp->prod_disabled || !prod_admit(vp)) {
            const LimbJob *f = (const LimbJob *)dF, *iv = (const LimbJob *)dI;
            const TileDesc *tf = (const TileDesc *)dFt, *ti = (const TileDesc *)dIt;
            hipError_t e2 = aloha::k::ntt_tile_batch(st, f, f + cnt2, vp->d_primes, 0, cnt2, lg, false, tf, nfu, tf + nfu, nfg, &launches, single_max, false);
            if (e2 == hipSuccess) { e2 = aloha::k::ew_batch(st, (const aloha::k::EwJob *)dE, nj, nn); launches++; }
            if (e2 == hipSuccess) e2 = aloha::k::ntt_tile_batch(st, iv, iv + nj, vp->d_primes, 0, nj, lg, true, ti, niu, ti + niu, nig, &launches, single_max, false);
            vp->stats.kernel_launches += launches;
            vp->stats.product_fallbacks++;
            return e2;
          }
          // a tag none of the slots [0, nj) still carries (slots a smaller launch skipped keep older tags); if launches of
          // different sizes have left all four behind, the slots are wiped first
          uint32_t used = 0;
          for (uint32_t i = 0; i < nj; i++) used |= 1u << vp->prod_tag[i];
          if (used == 15u) {
            const hipError_t em = hipMemsetAsync(vp->d_prod_work, 0, (size_t)nj * 3 * 8 << 14, st);
            if (em != hipSuccess) return em;
            used = 1u;
          }
          uint32_t tag = 1;
          while (used & (1u << tag)) tag = (tag + 1) & 3;
          for (uint32_t i = 0; i < nj; i++) vp->prod_tag[i] = (uint8_t)tag;
          const hipError_t e2 = aloha::k::poly_product_batch(st, (const aloha::k::ProdJob *)dj, vp->d_primes, nj, vp->d_prod_work, vp->d_prod_ctl,
                                                             tag, &launches);
          vp->stats.kernel_launches += launches;
          vp->prod_unchecked = true;
          if (e2 == hipSuccess) prod_launched(vp, st);
          return e2;
        });
        if (e != hipSuccess) s = fail(vp, ALOHA_ERR_DEVICE, "poly_product_batch: %s", hipGetErrorString(e));
      }
    }
    batch.clear(); aut_batch.clear(); ew_batch.clear(); dot_batch.clear(); prod_batch.clear(); pend_rw.clear();
    box_r_lo = box_r_hi = box_w_lo = box_w_hi = nullptr;
    dot_w_lo = dot_w_hi = dot_a_lo = dot_a_hi = dot_b_lo = dot_b_hi = nullptr;
    pend_kind = PK_NONE;
    return s;
  };
  // may a job that reads [a..), [b..) and writes [z..) join the pending batch of `kind`?
  auto can_join = [&](int kind, unsigned logn_, const u64 *a, const u64 *b, const u64 *z, size_t n) -> bool {
    if (pend_kind == PK_NONE) return true;
    if (pend_kind != kind || batch_logn != logn_) return false;
    if (!hits(z, n, box_w_lo, box_w_hi) && !hits(z, n, box_r_lo, box_r_hi) && !hits(a, n, box_w_lo, box_w_hi) &&
        !hits(b, n, box_w_lo, box_w_hi))
      return true;
    for (const auto &rw : pend_rw) {
      const u64 *r = rw.first, *w = rw.second;
      if (overlap(z, w, n) || (r && overlap(z, r, n))) return false;          // WAW / WAR
      if ((a && overlap(a, w, n)) || (b && overlap(b, w, n))) return false;    // RAW
    }
    return true;
  };
  // evaluate a lazy dot product into dst (the register's own storage by default)
  auto materialise = [&](unsigned r, u64 *dst, uint32_t n) -> aloha_status {
    if (lazy[r].empty()) return ALOHA_OK;
    aloha_status s = flush();                                 // its operands may be outputs of pending jobs
    if (s) return s;
    if (!dst) dst = vreg(vp, r);
    const u64 *init = nullptr;
    for (size_t t0 = 0; t0 < lazy[r].size(); t0 += aloha::k::kDotTerms) {
      aloha::k::DotArgs g{};
      g.nterms = (int)std::min<size_t>(aloha::k::kDotTerms, lazy[r].size() - t0);
      for (int t = 0; t < g.nterms; t++) {
        g.a[t] = lazy[r][t0 + t].first; g.b[t] = lazy[r][t0 + t].second; g.s[t] = lazy[r][t0 + t].s;
        g.sp[t] = (g.b[t] || !g.a[t]) ? 0 : aloha::nt::shoup(g.s[t], lazy_q[r]);
      }
      g.init = init; g.d = dst; g.m = modctx(lazy_q[r]);
      HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::dot(st, g, n); }));
      init = dst;
    }
    lazy[r].clear();
    return ALOHA_OK;
  };
  // Memory [dst, dst+n) is about to be written at next_pc-1: registers aliasing it either die
  // (dropped) or get their value copied out first; lazy dot products over it are evaluated.
  auto before_mem_write = [&](const u64 *dst, size_t n, uint32_t next_pc) -> aloha_status {
    for (unsigned r = 0; r < ALOHA_NVREG; r++) {
      for (const auto &t : lazy[r])
        if ((t.first && overlap(t.first, dst, n)) || (t.second && overlap(t.second, dst, n))) {
          if (dead_from(vp, next_pc, r)) lazy[r].clear();
          else { aloha_status s = materialise(r, nullptr, (uint32_t)n); if (s) return s; }
          break;
        }
      if (!alias[r] || !overlap(alias[r], dst, n)) continue;
      if (!dead_from(vp, next_pc, r)) {
        aloha_status s = flush();                            // the copy must see memory as of program order
        if (s) return s;
        {
          u64 *cd = vreg(vp, r); const u64 *cs = alias[r]; const size_t cb = n * 8;
          HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(cd, cs, cb, hipMemcpyDeviceToDevice, st); }));
        }
      }
      alias[r] = nullptr;
    }
    return ALOHA_OK;
  };
#define ALOHA_RUN_FAIL(status, ...) do { (void)flush(); return fail(vp, status, __VA_ARGS__); } while (0)
#define ALOHA_RUN_CK(expr) do { aloha_status s_ = (expr); if (s_) { (void)flush(); return s_; } } while (0)

  for (;; pc++) {
    if (pc >= vp->iram_depth) ALOHA_RUN_FAIL(ALOHA_ERR_NOBREAK, "no break before the end of the ROM");
    const Decoded d = decode_at(vp, pc);
    const unsigned funct6 = d.funct6, vs2 = d.vs2, vs1 = d.vs1, funct3 = d.funct3, vd = d.vd;
    const uint64_t imm = d.imm;
    vp->stats.instructions++;
    const bool is_cfg = funct6 == 0x04 || funct6 == 0x08 || funct6 == 0x0c || funct6 == 0x10 || funct6 == 0x00;
    unsigned logn = 0;
    uint32_t n = 0;
    const uint64_t q = vp->q;
    if (!is_cfg) {
      aloha_status s = vl_logn(vp->vl_bits, vp->max_logn, &logn);
      if (s) ALOHA_RUN_FAIL(s, "pc %u: VL = %llu bits", pc, (u64)vp->vl_bits);
      n = 1u << logn;
      if (funct6 != 0x03 && funct6 != 0x07 && funct6 != 0x19 && funct6 != 0x1d && !(q > (1ull << 59) && q < (1ull << 60)))
        ALOHA_RUN_FAIL(ALOHA_ERR_MODULUS, "pc %u: arithmetic with q = %#llx", pc, (u64)q);
    }
    // a following `vse vd, Z` this instruction may write to directly
    auto direct_store = [&](u64 **out) -> aloha_status {
      *out = nullptr;
      if (!fuse || pc + 1 >= vp->iram_depth) return ALOHA_OK;
      const Decoded nx = decode_at(vp, pc + 1);
      if (nx.funct6 != 0x07 || nx.vs1 != vd || !dead_from(vp, pc + 2, vd)) return ALOHA_OK;
      u64 *z = mem_rows(vp, nx.imm, false, csr, n / kLanes);
      if (!z) return ALOHA_OK;                               // let the vse itself report the address error
      *out = z;
      return ALOHA_OK;
    };

    // Lazy dot products: the accumulate pattern `vfqmul.vv t, a, b ; vfqadd.vv acc, acc, t` folds
    // (below); any other read of a lazy register evaluates it first, any overwrite drops it.
    const bool fold_add = fuse && funct6 == 0x05 && funct3 == 0 && ((vs1 ^ vs2) & 1) && (vd == vs1 || vd == vs2) &&
                          !lazy[vs1].empty() && !lazy[vs2].empty() && lazy_q[vs1] == q && lazy_q[vs2] == q &&
                          dead_from(vp, pc + 1, vd == vs1 ? vs2 : vs1);
    // ... and `vfqadd.vs / vfqsub.vs acc, acc, c` on a lazy sum appends a CONSTANT term (rounding offsets of the
    // mod-down, keyswitch.mem:81,:87): the sum stays lazy.
    const bool fold_const = fuse && (funct6 == 0x05 || funct6 == 0x09) && funct3 == 1 && !lazy[vs1].empty() &&
                            lazy_q[vs1] == q && (vd == vs1 || dead_from(vp, pc + 1, vs1)) &&
                            lazy[vs1].size() < (size_t)aloha::k::kDotTerms;
    if (!is_cfg && !fold_add && !fold_const) {
      const bool lazy_store = funct6 == 0x07 && !lazy[vs1].empty() && dead_from(vp, pc + 1, vs1);
      for (unsigned r = 0; r < ALOHA_NVREG; r++)
        if (!lazy[r].empty() && reads_reg(d, r) && !lazy_store) ALOHA_RUN_CK(materialise(r, nullptr, n));
    }

    switch (funct6) {
      case 0x10: {                                                 // break   expander.v:141-151
        const aloha_status sb = flush();
        if (!sb && vp->epoch == epoch0) finish_recording();      // (an eviction during the run may have freed what an earlier op points to)
        return sb;
      }
      case 0x00: break;                                            // nop
      case 0x04: {                                                 // vsetvl  :156-159
        unsigned l;
        aloha_status s = vl_logn(imm, vp->max_logn, &l);
        if (s) ALOHA_RUN_FAIL(s, "pc %u: vsetvl %#llx", pc, (u64)imm);
        if (imm != vp->vl_bits) {                                  // aliases and lazy values are sized by VL
          ALOHA_RUN_CK(flush());
          unsigned lo0 = 0;
          if (vl_logn(vp->vl_bits, vp->max_logn, &lo0) == ALOHA_OK)
            for (unsigned r = 0; r < ALOHA_NVREG; r++) ALOHA_RUN_CK(materialise(r, nullptr, 1u << lo0));
          for (unsigned r = 0; r < ALOHA_NVREG; r++) lazy[r].clear();
          for (unsigned r = 0; r < ALOHA_NVREG; r++)
            if (alias[r]) {
              unsigned lo = 0;
              if (vl_logn(vp->vl_bits, vp->max_logn, &lo) == ALOHA_OK)
              {
                u64 *cd = vreg(vp, r); const u64 *cs = alias[r]; const size_t cb = (size_t)8 << lo;
                HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(cd, cs, cb, hipMemcpyDeviceToDevice, st); }));
              }
              alias[r] = nullptr;
            }
        }
        vp->vl_bits = imm;
        break;
      }
      case 0x08:                                                   // vsetq   :160-163
        if (!modulus_ok(imm)) ALOHA_RUN_FAIL(ALOHA_ERR_MODULUS, "pc %u: vsetq %#llx", pc, (u64)imm);
        vp->q = imm;
        break;
      case 0x0c: vp->mu = imm; break;                              // vsetiq  :164-167 (RTL Barrett constant; unused here)
      case 0x01: case 0x05: case 0x09: case 0x0d: case 0x11: {     // vfqmul/vfqadd/vfqsub :178-373, vfqmod :374-395, vcp :396-417
        aloha::k::EwOp op;
        unsigned ra, rb = ALOHA_NVREG;
        uint64_t s = 0;
        if (funct6 == 0x0d || funct6 == 0x11) {
          ra = vs1; op = funct6 == 0x0d ? aloha::k::EW_MOD : aloha::k::EW_CP;
        } else if (funct3 == 0) {
          if (((vs1 ^ vs2) & 1) == 0) ALOHA_RUN_FAIL(ALOHA_ERR_BANK, "pc %u: .vv with v%u, v%u in one bank", pc, vs1, vs2);
          ra = vs1; rb = vs2;
          op = funct6 == 0x01 ? aloha::k::EW_MUL_VV : funct6 == 0x05 ? aloha::k::EW_ADD_VV : aloha::k::EW_SUB_VV;
        } else if (funct3 == 1) {
          ra = vs1; s = imm;
          op = funct6 == 0x01 ? aloha::k::EW_MUL_VS : funct6 == 0x05 ? aloha::k::EW_ADD_VS : aloha::k::EW_SUB_VS;
        } else if (funct3 == 2 && funct6 == 0x09) {
          ra = vs2; s = imm; op = aloha::k::EW_SUB_SV;
        } else ALOHA_RUN_FAIL(ALOHA_ERR_DECODE, "pc %u: funct3 %u on funct6 %#x", pc, funct3, funct6);
        s = s >= q ? s - q : s;                                    // the scalar is normalised like a lane input (modalu.sv:46)
        if (s >= q) ALOHA_RUN_FAIL(ALOHA_ERR_ARG, "pc %u: scalar operand %#llx >= 2q", pc, (u64)imm);
        if (fold_const) {                                          // sum +- c: one more term, no operand
          std::vector<LazyTerm> terms;
          terms.swap(lazy[vs1]);
          terms.push_back(LazyTerm{nullptr, nullptr, funct6 == 0x05 ? s : (s ? q - s : 0)});
          lazy[vd].swap(terms);
          lazy_q[vd] = q;
          alias[vd] = nullptr;
          break;
        }
        if (fold_add) {                                            // acc += t  with both sides lazy: concatenate the sums
          const unsigned t = vd == vs1 ? vs2 : vs1;
          lazy[vd].insert(lazy[vd].end(), lazy[t].begin(), lazy[t].end());
          lazy[t].clear();
          break;
        }
        const u64 *a = rd(ra), *b = rb < ALOHA_NVREG ? rd(rb) : nullptr;
        u64 *z = nullptr, *dst = vreg(vp, vd);
        ALOHA_RUN_CK(direct_store(&z));
        // element-wise in place (z == a or z == b) is safe; a partial overlap is not
        if (z && ((a != z && overlap(a, z, n)) || (b && b != z && overlap(b, z, n)))) z = nullptr;
        if (fuse && !z && op == aloha::k::EW_MUL_VS && alias[ra]) {
          // memory operand times a scalar kept in a register: a lazy term as well (base-conversion sums)
          lazy[vd].assign(1, LazyTerm{a, nullptr, s});
          lazy_q[vd] = q;
          alias[vd] = nullptr;
          break;
        }
        if (fuse && !z && op == aloha::k::EW_MUL_VV && alias[ra] && alias[rb]) {
          // product of two memory operands kept in a register: start a lazy dot product
          lazy[vd].assign(1, LazyTerm{a, b, 0});
          lazy_q[vd] = q;
          alias[vd] = nullptr;
          break;
        }
        lazy[vd].clear();
        if (z) {
          // deferred: joins the pending element-wise batch (one launch for many limbs)
          if (!can_join(PK_EW, logn, a, b, z, n)) ALOHA_RUN_CK(flush());
          ALOHA_RUN_CK(before_mem_write(z, n, pc + 2));
          if (!can_join(PK_EW, logn, a, b, z, n)) ALOHA_RUN_CK(flush());      // (before_mem_write may have changed nothing; cheap)
          aloha::k::EwJob jb{};
          jb.d = z; jb.a = a; jb.b = b ? b : a; jb.s = s; jb.s_shoup = aloha::nt::shoup(s, q); jb.m = modctx(q); jb.op = (int)op;
          ew_batch.push_back(jb);
          note_rw(a, z, n);
          if (b) note_rw(b, z, n);
          pend_kind = PK_EW; batch_logn = logn;
          alias[vd] = nullptr;
          pc++; vp->stats.instructions++;                          // the vse is done
          break;
        }
        ALOHA_RUN_CK(flush());
        {
          const u64 ssp = aloha::nt::shoup(s, q);
          const aloha::k::ModCtx mc = modctx(q);
          HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::ew(st, op, dst, a, b, s, ssp, mc, n); }));
        }
        alias[vd] = nullptr;
        break;
      }
      case 0x15: case 0x19: case 0x1d: case 0x21: {                // vaut / vroli  :460-501,:549-558; extensions: vautn (0x1d, NTT form), vautz (0x21, flipped zeros)
        const uint64_t k = (step + imm) & (n - 1);                 // scalar_iconn = csr_step + imm, kept mod N
                                                                   // (vxu_lane.sv:594, vxu_top_tb.sv:603)
        if (funct6 != 0x19 && !((step + imm) & 1))
          ALOHA_RUN_FAIL(ALOHA_ERR_AUT, "pc %u: vaut with even exponent %llu", pc, (u64)(step + imm));
        const u64 *in = rd(vs1);
        u64 *z = nullptr;
        ALOHA_RUN_CK(direct_store(&z));
        if (z && overlap(in, z, n)) z = nullptr;                   // a gather cannot run in place
        lazy[vd].clear();
        if (z) {
          // deferred: joins the pending permutation batch
          if (!can_join(PK_AUT, logn, in, nullptr, z, n)) ALOHA_RUN_CK(flush());
          ALOHA_RUN_CK(before_mem_write(z, n, pc + 2));
          if (!can_join(PK_AUT, logn, in, nullptr, z, n)) ALOHA_RUN_CK(flush());
          aloha::k::AutJob jb{};
          jb.out = z; jb.in = in; jb.q = q; jb.k = (uint32_t)k; jb.roll = funct6 == 0x19 ? 1 : funct6 == 0x1d ? 2 : funct6 == 0x21 ? 3 : 0;
          if (!jb.roll || jb.roll == 3) {
            uint64_t kinv = 1;                                     // k^-1 mod N by Newton iteration (k odd)
            for (int it = 0; it < 5; it++) kinv = kinv * (2 - k * kinv);
            jb.k_inv = (uint32_t)(kinv & (n - 1));
          }
          aut_batch.push_back(jb);
          note_rw(in, z, n);
          pend_kind = PK_AUT; batch_logn = logn;
          alias[vd] = nullptr;
          pc++; vp->stats.instructions++;
          break;
        }
        ALOHA_RUN_CK(flush());
        u64 *dst = vreg(vp, vd);
        const bool inplace = !z && in == dst;
        u64 *out = inplace ? vp->d_scratch : dst;
        if (funct6 == 0x15) {
          uint64_t kinv = 1;                                       // k^-1 mod N by Newton iteration (k odd)
          for (int it = 0; it < 5; it++) kinv = kinv * (2 - k * kinv);
          kinv &= (n - 1);
          HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::vaut(st, out, in, logn, q, k, kinv); }));
        } else if (funct6 == 0x21) {
          uint64_t kinv = 1;
          for (int it = 0; it < 5; it++) kinv = kinv * (2 - k * kinv);
          kinv &= (n - 1);
          HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::vautz(st, out, in, logn, q, k, kinv); }));
        } else if (funct6 == 0x1d) {
          HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::vautn(st, out, in, logn, k); }));
        } else {
          HIPCK(vp, issue(vp, [=]() -> hipError_t { vp->stats.kernel_launches++; return aloha::k::vroli(st, out, in, logn, k); }));
        }
        if (inplace) {
          const size_t cb = (size_t)n * 8;
          HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(dst, out, cb, hipMemcpyDeviceToDevice, st); }));
        }
        alias[vd] = nullptr;
        break;
      }
      case 0x02: case 0x06: {                                      // vntt / vintt  :418-459
        uint32_t slot;
        ALOHA_RUN_CK(get_slot(vp, q, logn, &slot));
        const bool inv = funct6 == 0x06;
        // The RTL ping-pongs between vs1 and vd once per stage (ntt_fsm.sv:80, vxu_lane.sv:315-318):
        // vs1 is UNDEFINED afterwards; here it keeps its value.
        u64 *z = nullptr;
        ALOHA_RUN_CK(direct_store(&z));
        // `vintt vd, .. ; vfqmul.vs vE, vd, c ; [vfqadd.vs vE, vE, a ;] vse vE, Z` with vd, vE dead afterwards: the scalar
        // rides in the transform's last stage, the constant on its canonical result (LimbJob::sc) -- one transform job
        // instead of a transform, a pass over the limb and a store.  Tile / single-pass kernels only (N >= 2^12).
        unsigned skip = 0;                                         // instructions consumed besides the vintt and the vse
        LimbJob scaled{};
        if (!z && inv && fuse && logn >= 12 && !(vp->flags & ALOHA_FLAG_RADIX2_ONLY) && pc + 2 < vp->iram_depth) {
          const Decoded m1 = decode_at(vp, pc + 1);
          if (m1.funct6 == 0x01 && m1.funct3 == 1 && m1.vs1 == vd && m1.imm < 2 * q) {
            unsigned e = m1.vd, at = pc + 2;
            uint64_t addc = 0;
            bool ok = (e == vd) || dead_from(vp, pc + 2, vd);
            Decoded nx = decode_at(vp, at);
            if (ok && nx.funct6 == 0x05 && nx.funct3 == 1 && nx.vs1 == e && nx.vd == e && nx.imm < 2 * q && at + 1 < vp->iram_depth) {
              addc = nx.imm >= q ? nx.imm - q : nx.imm;
              at++;
              nx = decode_at(vp, at);
            }
            // ... `vautz vF, vE, imm` before the store (N = 2^16): a zero at a sign-flipped target becomes q, on the canonical result
            uint32_t zk = 0, zkinv = 0;
            if (ok && nx.funct6 == 0x21 && nx.vs1 == e && logn == 16 && ((step + nx.imm) & 1) && at + 1 < vp->iram_depth &&
                (nx.vd == e || dead_from(vp, at + 1, e))) {
              const uint64_t kz = (step + nx.imm) & (n - 1);
              uint64_t kinv = 1;
              for (int it = 0; it < 5; it++) kinv = kinv * (2 - kz * kinv);
              zk = (uint32_t)kz; zkinv = (uint32_t)(kinv & (n - 1));
              e = nx.vd;
              at++;
              nx = decode_at(vp, at);
            }
            if (ok && nx.funct6 == 0x07 && nx.vs1 == e && dead_from(vp, at + 1, e)) {
              u64 *zz = mem_rows(vp, nx.imm, false, csr, n / kLanes);
              if (zz) {
                scaled.zk = zk; scaled.zkinv = zkinv;
                const uint64_t c = m1.imm >= q ? m1.imm - q : m1.imm;
                const auto &tt = vp->slots[slot].dev.tw;
                z = zz; skip = at - pc - 1;
                scaled.scaled = 1;
                scaled.sc[0] = aloha::nt::mulmod(tt.w1n_inv, c, q); scaled.sc[1] = aloha::nt::shoup(scaled.sc[0], q);
                scaled.sc[2] = aloha::nt::mulmod(tt.n_inv, c, q);   scaled.sc[3] = aloha::nt::shoup(scaled.sc[2], q);
                scaled.addc = addc;
              }
            }
          }
        }
        // `vntt t1, vA ; vntt t2, vB ; vfqmul.vv t3, t1, t2 ; vintt t4, t3 ; vse t4, Z` over memory operands vA, vB (loads may
        // stand anywhere among them), every temporary dead afterwards, N = 2^14: ONE job of the product kernel
        // (ntt_product.hip; BASELINE config 3) instead of two transforms, a product, an inverse transform and their stores.
        if (!z && !inv && fuse && logn == 14 && alias[vs1] && !(vp->flags & (ALOHA_FLAG_RADIX2_ONLY | ALOHA_FLAG_NTT_TWO_PASS)) &&
            !getenv("ALOHA_NO_PRODUCT")) {                         // (the variable: A/B timing switch)
          enum { S_NONE, S_MEM, S_NTT, S_PROD, S_RES };
          struct Sym { int kind; const u64 *a, *b; };
          Sym sym[ALOHA_NVREG];
          for (unsigned r = 0; r < ALOHA_NVREG; r++) sym[r] = Sym{(alias[r] && lazy[r].empty()) ? S_MEM : S_NONE, alias[r], nullptr};
          std::vector<std::pair<unsigned, u64 *>> loads;           // the window's own vle: (register, rows)
          u64 *zz = nullptr;
          const u64 *pa = nullptr, *pb = nullptr;
          uint32_t end = 0;
          bool ok = true;
          for (uint32_t i = pc; ok && !zz && i < pc + 10 && i < vp->iram_depth; i++) {
            const Decoded m = decode_at(vp, i);
            switch (m.funct6) {
              case 0x03: {
                u64 *p = mem_rows(vp, m.imm, true, csr, n / kLanes);
                if (!p) ok = false; else { sym[m.vd] = Sym{S_MEM, p, nullptr}; loads.emplace_back(m.vd, p); }
                break;
              }
              case 0x02: if (sym[m.vs1].kind == S_MEM) sym[m.vd] = Sym{S_NTT, sym[m.vs1].a, nullptr}; else ok = false; break;
              case 0x01:
                if (m.funct3 == 0 && ((m.vs1 ^ m.vs2) & 1) && sym[m.vs1].kind == S_NTT && sym[m.vs2].kind == S_NTT)
                  sym[m.vd] = Sym{S_PROD, sym[m.vs1].a, sym[m.vs2].a};
                else ok = false;
                break;
              case 0x06: if (sym[m.vs1].kind == S_PROD) sym[m.vd] = Sym{S_RES, sym[m.vs1].a, sym[m.vs1].b}; else ok = false; break;
              case 0x07:
                if (sym[m.vs1].kind == S_RES) { zz = mem_rows(vp, m.imm, false, csr, n / kLanes); pa = sym[m.vs1].a; pb = sym[m.vs1].b; end = i; ok = zz != nullptr; }
                else ok = false;
                break;
              default: ok = false;
            }
          }
          ok = ok && zz;
          for (unsigned r = 0; ok && r < ALOHA_NVREG; r++)
            if (sym[r].kind >= S_NTT && !dead_from(vp, end + 1, r)) ok = false;
          // The destination may not touch an operand.  (The one-launch kernel could run in place -- its columns are read before they
          // are written -- but a launch that gave up on a wait would then have destroyed its own input: with the operands intact the
          // caller can run the program again, and the handle does so by itself through the five-launch form from then on.)
          if (ok && (overlap(zz, pa, n) || overlap(zz, pb, n))) ok = false;
          if (ok) {
            if (!can_join(PK_PROD, logn, pa, pb, zz, n)) ALOHA_RUN_CK(flush());
            for (const auto &ld : loads) { lazy[ld.first].clear(); alias[ld.first] = ld.second; }
            for (unsigned r = 0; r < ALOHA_NVREG; r++)
              if (sym[r].kind >= S_NTT) { lazy[r].clear(); alias[r] = nullptr; }   // (dead: whatever they held is gone)
            ALOHA_RUN_CK(before_mem_write(zz, n, end + 1));
            if (!can_join(PK_PROD, logn, pa, pb, zz, n)) ALOHA_RUN_CK(flush());
            prod_batch.push_back(aloha::k::ProdJob{pa, pb, zz, slot, 0});
            note_rw(pa, zz, n);
            note_rw(pb, zz, n);
            pend_kind = PK_PROD; batch_logn = logn;
            vp->stats.instructions += end - pc;
            pc = end;
            break;
          }
        }
        const u64 *src = rd(vs1);
        if (z && src != z && overlap(src, z, n)) { z = nullptr; skip = 0; }
        lazy[vd].clear();
        if (z) {
          // fused `vle; vntt; vse`: join the pending transform batch unless it conflicts with it
          // (an in-place job, src == z, is fine: can_join sees only OTHER jobs' ranges)
          auto joinable = [&]() { return can_join(PK_NTT, logn, src, nullptr, z, n) && (batch.empty() || batch_inv == inv); };
          // forward transforms at N = 2^16 behind a batch of sums: the sums are held, not launched (staged batches, above)
          if (stage_ok && !inv && logn == 16 && pend_kind == PK_DOT && batch_logn == 16 && !dot_batch.empty() && held_sums.empty() &&
              held_ntt.empty() && held_ew.empty()) {
            held_sums.swap(dot_batch);
            reset_pending();
          }
          if (!joinable()) ALOHA_RUN_CK(flush());
          if (alias[vs1] == src && dead_from(vp, pc + 2 + skip, vs1)) alias[vs1] = nullptr;   // (undefined after the transform anyway)
          ALOHA_RUN_CK(before_mem_write(z, n, pc + 2 + skip));
          if (!joinable()) ALOHA_RUN_CK(flush());
          if (skip) {
            scaled.src = src; scaled.dst = z; scaled.prime = slot;
            batch.push_back(scaled);
            for (unsigned u = 1; u <= skip; u++) {                 // the registers the folded instructions would have written
              const Decoded mu_ = decode_at(vp, pc + u);
              lazy[mu_.vd].clear(); alias[mu_.vd] = nullptr;
            }
          } else {
            batch.push_back(LimbJob{src, z, slot, 0, {0, 0, 0, 0}, 0});
          }
          note_rw(src, z, n);
          pend_kind = PK_NTT; batch_inv = inv; batch_logn = logn;
          alias[vd] = nullptr;
          pc += 1 + skip; vp->stats.instructions += 1 + skip;
          break;
        }
        ALOHA_RUN_CK(flush());
        std::vector<LimbJob> job(1);
        job[0] = LimbJob{src, vreg(vp, vd), slot, 0, {0, 0, 0, 0}, 0};
        ALOHA_RUN_CK(run_transforms(vp, job, logn, inv));
        alias[vd] = nullptr;
        break;
      }
      case 0x03: {                                                 // vle  :647-661,:681-686
        u64 *p = mem_rows(vp, imm, true, csr, n / kLanes);
        if (!p) ALOHA_RUN_FAIL(ALOHA_ERR_ADDR, "pc %u: vle imm %#llx out of range", pc, (u64)imm);
        lazy[vd].clear();
        if (fuse) { alias[vd] = p; break; }                        // zero copy: alias until either side is overwritten
        {
          u64 *cd = vreg(vp, vd); const size_t cb = (size_t)n * 8;
          HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(cd, p, cb, hipMemcpyDeviceToDevice, st); }));
        }
        break;
      }
      case 0x07: {                                                 // vse (source register = vs1)  :502-523,:687-692
        u64 *p = mem_rows(vp, imm, false, csr, n / kLanes);
        if (!p) ALOHA_RUN_FAIL(ALOHA_ERR_ADDR, "pc %u: vse imm %#llx out of range", pc, (u64)imm);
        if (!lazy[vs1].empty()) {                                  // (only reached when vs1 is dead afterwards)
          bool clash = false;                                      // the sum must not read what it overwrites
          for (const auto &t : lazy[vs1]) clash |= (t.first && overlap(t.first, p, n)) || (t.second && overlap(t.second, p, n));
          if (!clash) {
            std::vector<LazyTerm> terms;
            terms.swap(lazy[vs1]);                                 // keep before_mem_write from evaluating it twice
            ALOHA_RUN_CK(before_mem_write(p, n, pc + 1));
            lazy[vs1].swap(terms);
            if (lazy[vs1].size() <= (size_t)aloha::k::kDotTerms) {
              // deferred: all such sums of a loop go out as ONE launch.  A pending dot job's writes are
              // its `d`; its reads are its operands: the new job may neither read nor overwrite a
              // pending destination, nor overwrite a pending operand.
              auto joinable = [&]() {
                if (pend_kind == PK_NONE) return true;
                if (pend_kind != PK_DOT || batch_logn != logn) return false;
                bool clear = !hits(p, n, dot_w_lo, dot_w_hi) && !hits(p, n, dot_a_lo, dot_a_hi) && !hits(p, n, dot_b_lo, dot_b_hi);
                for (const auto &t : lazy[vs1])
                  clear = clear && !hits(t.first, n, dot_w_lo, dot_w_hi) && !hits(t.second, n, dot_w_lo, dot_w_hi);   // (hits(nullptr) is false)
                if (clear) return true;
                for (const auto &g : dot_batch) {
                  if (overlap(p, g.d, n)) return false;
                  for (int t = 0; t < g.nterms; t++)
                    if ((g.a[t] && overlap(p, g.a[t], n)) || (g.b[t] && overlap(p, g.b[t], n))) return false;
                  for (const auto &t : lazy[vs1])
                    if ((t.first && overlap(t.first, g.d, n)) || (t.second && overlap(t.second, g.d, n))) return false;
                }
                return true;
              };
              // a sum behind a batch of forward transforms at N = 2^16: the transforms are held (staged batches, above)
              // sums behind a batch of element-wise products: the products are held (tensor product, above)
              if (stage_ok && pend_kind == PK_EW && batch_logn == logn && !ew_batch.empty() && held_ew.empty() && held_sums.empty() && held_ntt.empty()) {
                held_ew.swap(ew_batch);
                reset_pending();
              }
              if (stage_ok && logn == 16 && pend_kind == PK_NTT && !batch_inv && batch_logn == 16 && !batch.empty() && held_ntt.empty()) {
                held_ntt.swap(batch);
                reset_pending();
              }
              if (!joinable()) ALOHA_RUN_CK(flush());
              aloha::k::DotArgs g{};
              g.nterms = (int)lazy[vs1].size();
              for (int t = 0; t < g.nterms; t++) {
                g.a[t] = lazy[vs1][t].first; g.b[t] = lazy[vs1][t].second; g.s[t] = lazy[vs1][t].s;
                g.sp[t] = (g.b[t] || !g.a[t]) ? 0 : aloha::nt::shoup(g.s[t], lazy_q[vs1]);
              }
              g.init = nullptr; g.d = p; g.m = modctx(lazy_q[vs1]);
              dot_batch.push_back(g);
              auto grow = [&](const u64 *x, const u64 *&lo, const u64 *&hi) {
                if (!lo || x < lo) lo = x;
                if (!hi || x + n > hi) hi = x + n;
              };
              grow(p, dot_w_lo, dot_w_hi);
              for (int t = 0; t < g.nterms; t++) {
                if (g.a[t]) grow(g.a[t], dot_a_lo, dot_a_hi);
                if (g.b[t]) grow(g.b[t], dot_b_lo, dot_b_hi);
              }
              pend_kind = PK_DOT; batch_logn = logn;
              lazy[vs1].clear();
              break;
            }
            ALOHA_RUN_CK(materialise(vs1, p, n));                  // the dot product lands in memory directly
            break;
          }
          ALOHA_RUN_CK(materialise(vs1, nullptr, n));
        }
        ALOHA_RUN_CK(flush());
        const u64 *src = rd(vs1);
        if (src == p) break;                                       // storing an alias back onto itself
        if (overlap(src, p, n)) {                                  // shifted self-overlap: go through the register
          {
            u64 *cd = vp->d_scratch; const u64 *cs = src; const size_t cb = (size_t)n * 8;
            HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(cd, cs, cb, hipMemcpyDeviceToDevice, st); }));
          }
          src = vp->d_scratch;
        }
        ALOHA_RUN_CK(before_mem_write(p, n, pc + 1));
        {
          const size_t cb = (size_t)n * 8;
          HIPCK(vp, issue(vp, [=]() -> hipError_t { return hipMemcpyAsync(p, src, cb, hipMemcpyDeviceToDevice, st); }));
        }
        break;
      }
      default:                                                     // RTL: silent no-op bundle (:524-532)
        ALOHA_RUN_FAIL(ALOHA_ERR_DECODE, "pc %u: undefined funct6 %#x", pc, funct6);
    }
  }
#undef ALOHA_RUN_FAIL
#undef ALOHA_RUN_CK
}

extern "C" {

aloha_status aloha_vp_ntt_rows(aloha_vp *vp, int space, uint64_t dst_row, uint64_t src_row, uint32_t log_n,
                               const uint64_t *moduli, uint32_t count, int inverse) {
  if (!vp || !moduli) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  if (log_n < 10 || log_n > 16) return fail(vp, ALOHA_ERR_VL, "ntt_rows: log_n = %u", log_n);
  const uint64_t rows_per_limb = (1ull << log_n) / kLanes;
  const uint64_t bytes = (uint64_t)count * rows_per_limb * ALOHA_ROW_BYTES;
  u64 *src = space_ptr(vp, space, src_row, bytes), *dst = space_ptr(vp, space, dst_row, bytes);
  if (!src || !dst) return fail(vp, ALOHA_ERR_ADDR, "ntt_rows: %u limbs at rows %llu/%llu outside space %d", count, (u64)src_row, (u64)dst_row, space);
  if (src != dst && src < dst + bytes / 8 && dst < src + bytes / 8)      // limbs are transformed concurrently
    return fail(vp, ALOHA_ERR_ARG, "ntt_rows: source and destination ranges overlap without being equal");
  // steady state: identical arguments => reuse the resolved job list (no prime checks, no allocation)
  for (const RowsKey &k : vp->rows_cache)
    if (k.space == space && k.dst_row == dst_row && k.src_row == src_row && k.log_n == log_n && k.count == count &&
        memcmp(k.moduli.data(), moduli, (size_t)count * 8) == 0)
      return run_transforms(vp, k.jobs, log_n, inverse != 0);
  {
    std::set<uint64_t> fresh;
    for (uint32_t i = 0; i < count; i++)
      if ((i == 0 || moduli[i] != moduli[i - 1]) && !vp->slot_of.count(std::make_pair(moduli[i], (unsigned)log_n))) fresh.insert(moduli[i]);
    aloha_status s = make_room_for_twiddles(vp, fresh.size());
    if (s) return s;
  }
  std::vector<LimbJob> jobs(count);
  uint64_t last_q = 0;
  uint32_t last_slot = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (moduli[i] != last_q) {
      if (!modulus_ok(moduli[i])) return fail(vp, ALOHA_ERR_MODULUS, "ntt_rows: modulus[%u] = %#llx", i, (u64)moduli[i]);
      aloha_status s = get_slot(vp, moduli[i], log_n, &last_slot);
      if (s) return s;
      last_q = moduli[i];
    }
    jobs[i] = LimbJob{src + ((size_t)i << log_n), dst + ((size_t)i << log_n), last_slot, 0, {0, 0, 0, 0}, 0};
  }
  if (vp->rows_cache.size() >= kMaxPlans) vp->rows_cache.erase(vp->rows_cache.begin());
  RowsKey key;
  key.space = space; key.dst_row = dst_row; key.src_row = src_row; key.log_n = log_n; key.count = count;
  key.moduli.assign(moduli, moduli + count);
  key.jobs = jobs;
  vp->rows_cache.push_back(std::move(key));
  return run_transforms(vp, jobs, log_n, inverse != 0);
}


/* Synthetic residues straight into HBM: limb i of `count` limbs at row + i*(N/128) receives the first N
 * accepted 60-bit candidates of splitmix64(seed_base + first_index + i), accepted when < moduli[i]. */
aloha_status aloha_vp_fill_synthetic(aloha_vp *vp, int space, uint64_t row, uint32_t log_n, const uint64_t *moduli,
                                     uint32_t count, uint64_t seed_base, uint64_t first_index) {
  if (!vp || !moduli) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  if (log_n < 10 || log_n > 16) return fail(vp, ALOHA_ERR_VL, "fill_synthetic: log_n = %u", log_n);
  const uint64_t bytes = (uint64_t)count * ((1ull << log_n) / kLanes) * ALOHA_ROW_BYTES;
  u64 *dst = space_ptr(vp, space, row, bytes);
  if (!dst) return fail(vp, ALOHA_ERR_ADDR, "fill_synthetic: %u limbs at row %llu outside space %d", count, (u64)row, space);
  for (uint32_t i = 0; i < count; i++)
    if (moduli[i] < (1ull << 56) || moduli[i] > (1ull << 60))    // (an acceptance rate below 1/16 would make the rejection loop long)
      return fail(vp, ALOHA_ERR_MODULUS, "fill_synthetic: modulus[%u] = %#llx is outside [2^56, 2^60]", i, (u64)moduli[i]);
  if (!count) return ALOHA_OK;
  const void *d_mod = nullptr;
  aloha_status s = device_blob(vp, moduli, (size_t)count * 8, &d_mod);
  if (s) return s;
  HIPCK(vp, aloha::k::fill_splitmix(vp->stream, dst, (const u64 *)d_mod, count, 1u << log_n, seed_base, first_index));
  vp->stats.kernel_launches++;
  return ALOHA_OK;
}

/* Effective VALU clock (GHz) under an integer multiply-add load, measured now on this device: the figure that prices
 * the transforms' instruction count (bench.py: roofline.valu).  Runs the calibration kernel for about `ms` milliseconds
 * (clocks settle under load) and times its last launches. */
aloha_status aloha_vp_valu_clock(aloha_vp *vp, double ms, double *ghz) {
  if (!vp || !ghz || !(ms > 0)) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  hipDeviceProp_t prop;
  HIPCK(vp, hipGetDeviceProperties(&prop, vp->device));
  const uint32_t simds = (uint32_t)prop.multiProcessorCount * 4, iters = 4096;
  hipEvent_t e0, e1;
  HIPCK(vp, hipEventCreate(&e0));
  HIPCK(vp, hipEventCreate(&e1));
  double best_ms = 0;
  float total = 0;
  for (int rep = 0; rep < 1000 && total < ms; rep++) {
    HIPCK(vp, hipEventRecord(e0, vp->stream));
    for (int k = 0; k < 4; k++) HIPCK(vp, aloha::k::valu_clock(vp->stream, vp->d_scratch, simds, iters));
    HIPCK(vp, hipEventRecord(e1, vp->stream));
    HIPCK(vp, hipEventSynchronize(e1));
    float t = 0;
    HIPCK(vp, hipEventElapsedTime(&t, e0, e1));
    total += t;
    best_ms = t / 4;                                     // the LAST block: the settled clock
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double cycles = 8.0 * 32.0 * iters * 4.0;        // per SIMD and launch (see valu_clock_kernel)
  *ghz = cycles / (best_ms * 1e-3) / 1e9;
  return ALOHA_OK;
}

}  // extern "C" (closed for two file-local helpers)

// Tables of the encoder block for ring size N = 2^log_n (n = N/2): w[k] = exp(-2 pi i k / n),
// twist[k] = exp(-i pi k / N) in long double, and the slot map of the rotation group generated by 3.
static aloha_status enc_tables_for(aloha_vp *vp, uint32_t log_n, const aloha::k::EncodeTables **out) {
  const size_t N = (size_t)1 << log_n, n = N / 2;
  auto it = vp->enc_tables.find(log_n);
  if (it == vp->enc_tables.end()) {
    std::vector<double> host(n + 2 * n + n / 2 + 8);        // w (n/2 complex) | twist (n complex) | src (n u32)
    double *w = host.data(), *tw = w + n;
    uint32_t *src = reinterpret_cast<uint32_t *>(tw + 2 * n);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (size_t k = 0; k < n / 2; k++) {
      const long double a = -2.0L * pi * (long double)k / (long double)n;
      w[2 * k] = (double)cosl(a); w[2 * k + 1] = (double)sinl(a);
    }
    for (size_t k = 0; k < n; k++) {
      const long double a = -pi * (long double)k / (long double)N;
      tw[2 * k] = (double)cosl(a); tw[2 * k + 1] = (double)sinl(a);
    }
    uint64_t e = 1;
    for (size_t j = 0; j < n; j++) {
      if (e % 4 == 1) src[(e - 1) / 4] = (uint32_t)j;
      else src[(2 * N - e - 1) / 4] = (uint32_t)j | 0x80000000u;
      e = e * 3 % (2 * N);
    }
    aloha_vp::EncTab tab;
    HIPCK(vp, hipMalloc(&tab.alloc, host.size() * 8));
    HIPCK(vp, hipMemcpyAsync(tab.alloc, host.data(), host.size() * 8, hipMemcpyHostToDevice, vp->stream));
    HIPCK(vp, hipStreamSynchronize(vp->stream));
    tab.t.w = reinterpret_cast<const double *>(tab.alloc);
    tab.t.twist = tab.t.w + n;
    tab.t.src = reinterpret_cast<const uint32_t *>(tab.t.twist + 2 * n);
    it = vp->enc_tables.emplace(log_n, tab).first;
  }
  *out = &it->second.t;
  return ALOHA_OK;
}

static aloha_status grow(aloha_vp *vp, void **buf, size_t *have, size_t need) {
  if (need <= *have) return ALOHA_OK;
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr; *have = 0;
  HIPCK(vp, hipMalloc(buf, need));
  *have = need;
  return ALOHA_OK;
}

extern "C" {

// CKKS encoder block: see encode_kernels.hip / include/aloha_vp.h.  Tables for ring size N (n = N/2):
// w[k] = exp(-2 pi i k / n), twist[k] = exp(-i pi k / N) in long double, and the slot map of the
// rotation group generated by 3 (run_rotate's k = 3^step, top_noaxilite_tb.sv:531).
aloha_status aloha_vp_encode(aloha_vp *vp, int space, uint64_t dst_row, const double *host_re_im, uint32_t nplain,
                             uint32_t log_n, const uint64_t *moduli, uint32_t nlimbs, double scale) {
  if (!vp || !host_re_im || !moduli || !nplain || !nlimbs || !(scale > 0)) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  if (log_n < 10 || log_n > 16) return fail(vp, ALOHA_ERR_VL, "encode: log_n = %u", log_n);
  for (uint32_t l = 0; l < nlimbs; l++)
    if (!modulus_ok(moduli[l])) return fail(vp, ALOHA_ERR_MODULUS, "encode: modulus[%u] = %#llx", l, (u64)moduli[l]);
  const size_t N = (size_t)1 << log_n, n = N / 2;
  const uint64_t bytes = (uint64_t)nplain * nlimbs * N * 8;
  u64 *dst = space_ptr(vp, space, dst_row, bytes);
  if (!dst) return fail(vp, ALOHA_ERR_ADDR, "encode: %u x %u limbs at row %llu outside space %d", nplain, nlimbs, (u64)dst_row, space);
  const aloha::k::EncodeTables *tabs = nullptr;
  { aloha_status ts = enc_tables_for(vp, log_n, &tabs); if (ts) return ts; }
  const size_t in_bytes = (size_t)nplain * N * 8, work_bytes = log_n > 14 ? (size_t)nplain * n * 16 : 0;
  { aloha_status gs = grow(vp, &vp->d_enc_in, &vp->enc_in_bytes, in_bytes); if (gs) return gs; }
  { aloha_status gs = grow(vp, &vp->d_enc_work, &vp->enc_work_bytes, work_bytes); if (gs) return gs; }
  const void *d_moduli = nullptr;
  aloha_status s = device_blob(vp, moduli, (size_t)nlimbs * 8, &d_moduli);
  if (s) return s;
  HIPCK(vp, hipMemcpyAsync(vp->d_enc_in, host_re_im, in_bytes, hipMemcpyHostToDevice, vp->stream));
  uint64_t launches = 0;
  hipError_t e = aloha::k::encode(vp->stream, reinterpret_cast<const double *>(vp->d_enc_in), *tabs,
                                  vp->d_enc_work, dst, reinterpret_cast<const u64 *>(d_moduli), nlimbs, nplain, log_n,
                                  scale, &launches);
  vp->stats.kernel_launches += launches;
  if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "encode launch failed: %s", hipGetErrorString(e));
  HIPCK(vp, hipStreamSynchronize(vp->stream));         // the host buffer is borrowed for the call only
  return ALOHA_OK;
}

// The inverse of aloha_vp_encode from ONE coefficient-form limb per plaintext (the reference has no
// decoder block; this closes the loop for users and tests).
aloha_status aloha_vp_decode(aloha_vp *vp, int space, uint64_t src_row, uint64_t limb_stride, double *host_re_im,
                             uint32_t nplain, uint32_t log_n, uint64_t modulus, double scale) {
  if (!vp || !host_re_im || !nplain || !(scale > 0)) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  if (log_n < 10 || log_n > 16) return fail(vp, ALOHA_ERR_VL, "decode: log_n = %u", log_n);
  if (!modulus_ok(modulus)) return fail(vp, ALOHA_ERR_MODULUS, "decode: modulus = %#llx", (u64)modulus);
  const size_t N = (size_t)1 << log_n, n = N / 2;
  if (limb_stride == 0 && nplain > 1) return ALOHA_ERR_ARG;
  const uint64_t span = ((uint64_t)(nplain - 1) * limb_stride + 1) * N * 8;
  const u64 *src = space_ptr(vp, space, src_row, span);
  if (!src) return fail(vp, ALOHA_ERR_ADDR, "decode: %u limbs (stride %llu) at row %llu outside space %d", nplain, (u64)limb_stride, (u64)src_row, space);
  const aloha::k::EncodeTables *tabs = nullptr;
  { aloha_status ts = enc_tables_for(vp, log_n, &tabs); if (ts) return ts; }
  { aloha_status gs = grow(vp, &vp->d_enc_in, &vp->enc_in_bytes, (size_t)nplain * N * 8); if (gs) return gs; }
  { aloha_status gs = grow(vp, &vp->d_enc_work, &vp->enc_work_bytes, (size_t)nplain * n * 16); if (gs) return gs; }
  uint64_t launches = 0;
  hipError_t e = aloha::k::decode(vp->stream, src, (size_t)limb_stride * N, *tabs, vp->d_enc_work,
                                  reinterpret_cast<double *>(vp->d_enc_in), modulus, nplain, log_n, scale, &launches);
  vp->stats.kernel_launches += launches;
  if (e != hipSuccess) return fail(vp, ALOHA_ERR_DEVICE, "decode launch failed: %s", hipGetErrorString(e));
  HIPCK(vp, hipMemcpyAsync(host_re_im, vp->d_enc_in, (size_t)nplain * N * 8, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return ALOHA_OK;
}

aloha_status aloha_find_psi(uint64_t q, uint32_t log_n, uint64_t *psi) {
  if (!psi) return ALOHA_ERR_ARG;
  if (!modulus_ok(q)) return ALOHA_ERR_MODULUS;
  return aloha::nt::find_psi(q, log_n, psi) ? ALOHA_OK : ALOHA_ERR_ROOT;
}

uint64_t aloha_nth_prime(uint32_t log_n, uint32_t idx) {
  if (log_n < 1 || log_n > 16) return 0;
  return aloha::nt::nth_prime(log_n, idx);
}

aloha_status aloha_vp_get_stats(const aloha_vp *vp, aloha_vp_stats *out) {
  if (!vp || !out) return ALOHA_ERR_ARG;
  *out = vp->stats;
  return ALOHA_OK;
}

aloha_status aloha_vp_peek_vreg(aloha_vp *vp, uint32_t reg, uint64_t *dst, uint64_t n) {
  if (!vp || !dst || reg >= ALOHA_NVREG || n > (1ull << vp->max_logn)) return ALOHA_ERR_ARG;
  DeviceGuard dev_guard(vp);
  HIPCK(vp, hipMemcpyAsync(dst, vreg(vp, reg), n * 8, hipMemcpyDeviceToHost, vp->stream));
  HIPCK(vp, hipStreamSynchronize(vp->stream));
  return ALOHA_OK;
}

}  // extern "C"
