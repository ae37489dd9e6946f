// occ_gibbs.hip -- host side of the engine behind include/occ_gibbs.h: device-resident problem and
// chain state, kernel sequencing (eager and hipGraph replay), state access, error mapping.
// HIP runtime only (no PyTorch, no BLAS/solver libraries).
#include "../../include/occ_gibbs.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "occ_comm.hpp"
#include "occ_tiles.hpp"
#include "occ_rsr.hpp"

using namespace occ;

namespace {

thread_local std::string g_create_error;

enum Kind { K_OMEGA_B = 0, K_NOISE, K_ETA_INIT, K_MINRES, K_BETA_PARTIAL, K_OMEGA_A, K_ALPHA_DRAW, K_Z_OB, K_ITER, K_GATE /* internal: not a profiled kind */,
            K_RSR_GRAM, K_RSR_SOLVE, K_RSR_ETA_BETA /* reduced-rank model */ };
static_assert(K_ITER + 1 == OCC_N_KERNEL_KINDS, "kernel kinds out of sync with the header");

// ---- streams are a PROCESS-WIDE resource, not an engine's ------------------------------------------------------------
// An MI355X gives a device 24 hardware queue slots for compute (KFD topology: num_cp_queues 24), shared by everything
// that runs on it.  The HIP runtime multiplexes ordinary streams onto a few queues per priority, but every CU-masked
// stream gets a hardware queue OF ITS OWN (tools/queue_probe.hip: with 12 live pairs of masked streams -- 24 queues plus
// the runtime's own -- the scheduler starts to time-slice the queues in quanta of about 10 ms; a kernel that polls for a
// word only a kernel on ANOTHER queue can set then waits a quantum or more whenever that queue is not mapped: 10.1 ms and
// 18 ms+ seen where a hand-over takes 30 us).  Round 2 gave every engine its own two masked streams, so a process that
// kept a dozen samplers alive ran every device-side hand-over of a new engine into its time-out -- and the probe at
// creation could not see it, because a fresh queue starts out mapped and loses its slot later, when other queues
// get work.  Hence:
//   * one (main, side) pair per (process, device, CU partition), shared by every engine that wants that partition; when
//     the last of them closes the pair stays, IDLE, for the next taker of the partition (release_pair) -- idle pairs are
//     destroyed only to make room under the cap (acquire_pair) and at process exit (pool_at_exit);
//   * at most MAX_MASKED_PAIRS masked pairs alive per device; an engine that would need one more takes the unmasked
//     pair and hands over through events -- decided by counting, not by probing;
//   * engines of one device run their calls ONE AFTER THE OTHER (DeviceSlot::busy, taken by every entry point): they
//     share streams, and two fused engines at once take each other's CUs anyway (their barriers need residency);
//   * every creation / destruction of a pair bumps g_stream_gen; an engine whose last stream probe is older than that
//     asks again at its next occ_run / occ_step (the mapping can change after creation).
struct StreamPair {
    int device = 0;
    std::vector<uint32_t> m_main, m_side;  // empty: the unmasked pair (priority streams)
    hipStream_t main = nullptr, side = nullptr;
    int refs = 0;
    bool dead = false;  // a host wait on one of its streams ran into its deadline, or the process is exiting: never handed out again
};
struct DeviceSlot {
    std::recursive_mutex busy;        // held for the length of every entry point that touches the device through a handle
    std::vector<StreamPair *> pairs;  // guarded by g_pool_mu
};
constexpr int MAX_MASKED_PAIRS = 4;  // per device: 8 of its 24 hardware queues
std::mutex g_pool_mu;
std::map<int, DeviceSlot *> g_slots;  // never freed: a handle's lease may outlive every pair
std::atomic<unsigned long long> g_stream_gen{1};
std::atomic<long long> g_pairs_evicted{0};  // idle masked pairs destroyed to make room under the cap (occ_stats)

DeviceSlot &device_slot(int device)
{
    std::lock_guard<std::mutex> g(g_pool_mu);
    DeviceSlot *&sl = g_slots[device];
    if (!sl) sl = new DeviceSlot();
    return *sl;
}
using DeviceLease = std::unique_lock<std::recursive_mutex>;
inline DeviceLease lease_device(int device) { return DeviceLease(device_slot(device).busy); }

}  // namespace

struct occ_sampler {
    int device = 0;
    StreamPair *pair = nullptr;    // the streams below belong to this pooled pair (shared with other engines of the device)
    hipStream_t stream = nullptr;  // main: eta_init -> minres ... -> beta -> z_ob
    hipStream_t side = nullptr;    // side: omega_a -> alpha_draw -> noise(t+1), forked/joined inside the graph
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool side_enabled = true;   // false: OCC_NO_SIDE_STREAM diagnostic
    // fork/join as event wait/record nodes inside the two graphs: two graph launches per iteration and no
    // graph -> kernel -> graph transitions on the critical path (8-13 us each on MI355X / ROCm 7.2)
    bool event_nodes = true;
    Ctx ctx{};               // host copy of the descriptor
    Ctx *ctx_dev = nullptr;  // the copy kernels read
    KryArgs kry{};           // by-value argument block of k_minres
    IterArgs iter{};         // ... and of k_iter
    // reduced-rank model (LogitRSRGibbs): rsr.m > 0.  k_rsr_gram / k_rsr_solve / k_rsr_eta_beta stand where k_iter is;
    // scheduled like the fused ICAR iteration (two streams with flag hand-overs, or one linear graph of GRAPH_SEQ iterations).
    RsrArgs rsr{};
    std::vector<double> rsr_K_host;  // n x m, for theta -> eta on the host (start values, set_state)
    // fused iteration (occ_iter.hpp): k_iter + k_z_ob on one stream, the eta solve persistent inside k_iter;
    // otherwise one launch per MINRES step on the main stream and omega_a / alpha / noise on the side stream
    bool persistent = false;
    int iter_window = 8;     // neighbour window of k_iter: 8 (two workgroups per CU) or 16 (rows of 9-16 off-diagonals, one per CU)
    bool xcd_local = false;  // k_iter<8, 1>: one XCD per chain, exchange through that XCD's L2 (occ_iter.hpp)
    bool xl_candidate = false, fused_fallback = false;
    // k_tiles (occ_tiles.hpp): the persistent solve for problems too large for k_iter -- tiles of 256 sites with their vectors
    // in LDS, T tiles per workgroup, G workgroups per chain in eight bands of B (one per XCD).  tiles_layout: the problem has
    // that shape (256-thread blocks, the MINRES sums added in groups of T blocks: the launch-per-step kernels follow the
    // same order, KryArgs::group_T); tiles: k_tiles is what K_ITER launches.
    bool tiles_layout = false, tiles = false;
    int tiles_T = 1, tiles_G = 0, tiles_B = 0;
    bool any_fits = false;   // the any-placement form fits the main stream's CUs (arithmetic; the residency probe decides)
    int nbg_any = 0;         // its workgroups per chain
    int tpb_plain = 256;     // threads per block of the launch-per-step path when no fused form applies
    int iter_flags_extra = 0;  // OR-ed into k_iter's flags (2: residency probe)
    bool generic = false;      // more than 8 occupancy or detection covariates: the P = 0 / Q = 0 kernels (run-time p, q), launch-per-step path
    bool beta_split = false;   // beta drawn by k_beta_draw (one wave per chain) in front of k_z_ob: many blocks, launch-per-step path
    int zob_debug = 0;         // OCC_DEBUG_ZOB_SKIP (timing experiments with occ_profile only): 8 = no z update, 16 = no omega_b draw
    bool device_timeout = false;  // the last error was a device-side wait that gave up (not a HIP API failure)
    // state of every chain at the start of the running occ_run / occ_step (fused engines only): what the call is re-run from
    double *snap_eta = nullptr;
    uint8_t *snap_z = nullptr;
    double2 *snap_x = nullptr;
    double *snap_theta = nullptr;  // reduced-rank model: the basis coefficients
    std::vector<ChainScalars> snap_sc;
    // fixed problem arrays on the device, in upload order: what a group broadcasts from its root (occ_create_group /
    // occ_create_distributed); defer_fixed: allocate only, the bytes arrive by broadcast
    std::vector<std::pair<void *, size_t>> fixed_list;
    std::vector<const char *> fixed_names;  // ... and what each one is (error messages of the after-broadcast check)
    bool defer_fixed = false;
    std::string group_transport = "none";
    Inject *inj_dev = nullptr;  // injected variates of the occ_cond_* entry points
    double *inj_u = nullptr;    // [n] uniforms of occ_cond_z
    int snap_parity = 0;
    int64_t fused_fallbacks = 0;  // occ_run calls that were re-run on the launch-per-step path after a device-side time-out
    int xl_wide = 0;         // ... with 512-thread workgroups (a chain needs more waves than its XCD's main-stream SIMDs): 1 = a scalar
                             // wave beside seven site waves, 2 = eight site waves, the first one leads
    int xl_nbg = 0;          // workgroups per chain of the XCD-local form
    int xl_per_cu = 1, xl_main = 0;  // its workgroups per CU; CUs of the main stream it wants (0: no partition)
    int xl_per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... per XCD, when the XCDs that host a chain get more than the others (first entry 0: evenly)
    int main_hot_cus = 0;    // CUs the main stream's mask holds on XCD 0 (a chain's XCD)
    bool streams_serialised = false;  // the last stream probe found the two streams NOT running beside each other: hand-overs by events
    unsigned *probe_w = nullptr;      // the stream probe's words
    unsigned long long probe_gen = 0; // g_stream_gen at the last stream probe
    int64_t stream_probes = 0, repromotions = 0;
    unsigned *sync_buf = nullptr;     // the hand-over counters (Ctx::sync points here while they are in use)
    // What creation decided -- the form of the fused kernel, the CU partition, device-side hand-overs -- kept so that an
    // engine that had to leave it at run time (fallback_to_launch_per_step) can come back (try_repromote)
    struct Preferred {
        bool valid = false, persistent = false, xcd_local = false, flag_sync = false, tiles = false;
        int share_on = 0, main_cus = 0;
        std::vector<uint32_t> m_main, m_side;
    } pref;
    bool demoted = false;             // running without device-side waits after one of them gave up
    int promote_wait = 0, promote_backoff = 1;  // calls until the next attempt to come back; doubled after a failed one
    std::string pair_note;            // why the engine did not get the masked pair it wanted (empty: it did, or wanted none)
    // The chains' scalars as the last occ_run / occ_step read them at its end (round 3: a call of a dozen iterations was mostly
    // host round trips -- ~170 us per occ_run beyond its iterations).  ONE flag says whether the device still holds exactly
    // these: `mirror_ok`, set at the clean end of occ_run / occ_step and dropped by open_window (the next call's kernels
    // change the scalars) and by every other writer of Ctx::sc (write_scalars, occ_profile).  Round 3 had five flags
    // (snap_fresh, sc_host_valid, window_open, clock_fresh, clean_exit) for what is now this one and `clean_exit`.
    std::vector<ChainScalars> mirror;
    bool mirror_ok = false;
    bool clean_exit = false;  // the last call through open_window ended cleanly: the claim / abort words are as the kernels leave them
    // Page-locked staging for what crosses PCIe on every call (an asynchronous copy to or from pageable memory is staged by
    // the runtime and, device to host, waits for the stream: two serial round trips at the end of every occ_run):
    // pin_sc = [2][C] ChainScalars (0: open_window's upload, 1: read_scalars' read-back), pin_rec = the recorded rows.
    ChainScalars *pin_sc = nullptr;
    double *pin_rec = nullptr;
    size_t pin_rec_cap = 0;
    bool marks_done = false;       // ev1 and the copy of the records are enqueued behind the call's last batch
    bool debug_close_window = false;  // tests (occ_set_state "debug_close_window"): the NEXT call's window is opened with zero iterations on the device
    bool wedged = false;           // a host wait ran into its deadline: the streams never drained; nothing of this engine is freed or reused
    int probe_wait = 0, probe_backoff = 1;  // calls until the stream probe is asked again after a negative answer
    int share_cum[2][9] = {};  // cumulative CUs of the main / side stream's mask over the XCDs (Ctx::share_on)
    int main_cus = 0;        // > 0: the main stream is restricted to this many CUs, the side stream to the others
    // stream hand-overs by device-side sequence counters (Ctx::sync) instead of event nodes: only with the CU
    // partition.  launch_sync = false makes launch_kind() launch kernels that neither wait nor publish
    // (prologue, timing loops).
    bool flag_sync = false, launch_sync = true;
    int graph_parity = 0;    // fused mode: sequence parity the captured pair of iterations starts with
    int tpb = 256;
    std::vector<void *> allocs;
    std::string err;
    int launch_rc = 0;       // first failed kernel launch since the last take_launch_rc()
    std::string launch_err;
    // launch-sequence ("slot") parity: the kernels of the next sequence read ChainScalars::ctl[parity]
    int parity = 0;
    // start values / state were just set by the host: omega_b and the noise of the current iteration
    // have to be produced stand-alone before the first sequence
    bool need_prologue = true;
    // graph replay.  Every graph is a LINEAR chain replayed on one of the engine's own two streams:
    // head[e] = k_iter (or k_eta_init, cap + 3 k_minres launches, k_beta_partial), wait for the
    // side chain, k_z_ob, record -- on the main stream; tail[e] = wait for the previous k_z_ob, k_omega_a,
    // k_alpha_draw, k_noise, record -- on the side stream.  (Graphs with parallel branches get
    // runtime-internal streams at every instantiation; on ROCm 7.2 the second or third such instantiation
    // shares a hardware queue with the launch stream and every kernel then runs 2-5x slower.  Linear graphs
    // can be re-instantiated freely, which the adaptive Krylov cap of the k_minres path needs.)
    hipGraph_t head_graph[2] = {nullptr, nullptr}, tail_graph[2] = {nullptr, nullptr};
    hipGraphExec_t head[2] = {nullptr, nullptr}, tail[2] = {nullptr, nullptr};
    hipEvent_t ev_z[2] = {nullptr, nullptr}, ev_side[2] = {nullptr, nullptr};
    int krylov_cap = 0;
    // statistics
    int64_t iterations = 0, graph_launches = 0, eager_iterations = 0, stalls = 0;
    int krylov_last = 0;
    double last_run_ms = 0.0;
    double profile_minres_iterations = 0.0;
    double profile_iter_dispatch_us = 0.0;
    hipEvent_t ext_ev0 = nullptr, ext_ev1 = nullptr;  // occ_profile: start / stop events of the next k_iter dispatch (hipExtLaunchKernel)
    int calib_max = 0;
    unsigned long long seen_tot = 0, seen_sq = 0, seen_solves = 0;  // counters at the last cap decision
    // record buffer (alpha | beta | tau rows of the current occ_run), kept between runs
    double *rec_buf = nullptr;
    size_t rec_cap = 0;
    // host mirrors
    std::vector<int32_t> site_id, site_ptr;
    std::vector<uint8_t> obs_site;
};

namespace {

// launch sequences (iterations) per captured graph on the paths that need no host decision between iterations:
// even, so that the sequence parity is the same at every replay; a graph boundary costs several microseconds
constexpr int TILES_GB_DEFAULT = 1;  // tiles of a k_tiles workgroup whose gathers are in flight together (T = 4, diagonal form)
constexpr int GRAPH_SEQ = 2;  // (16 per graph measured the same: the boundary between two replays is not what costs)

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            s->err = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
            return OCC_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

// ---- host waits have a deadline and name themselves ------------------------------------------------------------------
// Every device-side wait of the engine is bounded (SYNC_SPIN_LIMIT, ITER_SPIN_LIMIT), so a host wait on one of its streams
// that outlasts seconds is not the kernels' doing (round 3 recorded two such stops, one inside occ_create and one inside
// occ_run, with nothing but Python frames to go by: DESIGN 7).  The engine therefore never blocks inside
// hipStreamSynchronize / hipDeviceSynchronize: it polls hipStreamQuery against a deadline (OCC_HOST_WAIT_S, default 20 s)
// and a wait that runs into it returns OCC_E_HIP naming the call site and the engine's host-side state; the engine is then
// `wedged` -- its streams hold work that never completed, so nothing of it is freed, destroyed or handed to another engine.
static std::string host_state(const occ_sampler *s)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "device %d, %d chains, n %d; path: persistent %d xcd_local %d tiles %d rsr_m %d; hand-overs: flag_sync %d side_enabled %d "
                  "event_nodes %d demoted %d streams_serialised %d; main_cus %d (%s pair); parity %d graph_parity %d snap_parity %d; "
                  "mirror_ok %d clean_exit %d need_prologue %d; graph launches %lld, eager iterations %lld, fallbacks %lld",
                  s->device, s->ctx.C, s->ctx.n, (int)s->persistent, (int)s->xcd_local, (int)s->tiles, s->rsr.m, (int)s->flag_sync, (int)s->side_enabled,
                  (int)s->event_nodes, (int)s->demoted, (int)s->streams_serialised, s->main_cus, (s->pair && !s->pair->m_main.empty()) ? "CU-masked" : "plain",
                  s->parity, s->graph_parity, s->snap_parity, (int)s->mirror_ok, (int)s->clean_exit, (int)s->need_prologue,
                  (long long)s->graph_launches, (long long)s->eager_iterations, (long long)s->fused_fallbacks);
    return buf;
}
static double host_wait_limit_s()
{
    const char *e = std::getenv("OCC_HOST_WAIT_S");
    const double v = e ? std::atof(e) : 20.0;
    return v > 0.0 ? v : 20.0;
}
// hipSuccess when `st` has drained; hipErrorNotReady after the deadline (*late = true); any other error as the runtime gave it
static hipError_t poll_stream(hipStream_t st, bool *late)
{
    *late = false;
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = host_wait_limit_s();
    for (unsigned polls = 1;; ++polls) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();  // (not ready is not an error to keep)
        if ((polls & 63u) == 0u) {
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (sec > limit) { *late = true; return hipErrorNotReady; }
            if (sec > 0.002) std::this_thread::yield();  // (a long wait: the host thread need not own a core)
        }
    }
}
static int wait_on(occ_sampler *s, hipStream_t st, const char *func, int line)
{
    const std::string what = std::string(func) + " (occ_gibbs.hip:" + std::to_string(line) + ")";
    if (!st) return OCC_OK;
    bool late = false;
    const hipError_t e = poll_stream(st, &late);
    if (e == hipSuccess) return OCC_OK;
    if (late) {
        s->wedged = true;
        if (s->pair) {
            std::lock_guard<std::mutex> g(g_pool_mu);  // (the pool reads the mark under its lock)
            s->pair->dead = true;
        }
        char lim[64];
        std::snprintf(lim, sizeof(lim), "%.0f s", host_wait_limit_s());
        s->err = std::string("host wait `") + what + "`: the " + (st == s->side ? "side" : "main") + " stream did not drain within " + lim +
                 " (every device-side wait of the engine is bounded far below that: the queue is not being served); " + host_state(s);
    } else {
        s->err = std::string("host wait `") + what + "`: " + hipGetErrorString(e);
    }
    return OCC_E_HIP;
}
#define WAIT_TRY(st)                                           \
    do {                                                       \
        const int w_ = wait_on(s, (st), __func__, __LINE__);   \
        if (w_ != OCC_OK) return w_;                           \
    } while (0)

// Copies and fills are ordered on the engine's OWN stream, never on the legacy default stream: a default-stream operation
// synchronises implicitly with every blocking stream of the device (the CU-masked streams are blocking), which
// invalidates a stream capture that another host thread -- another engine on the same device -- has open at that moment.
static hipError_t drain_for_copy(occ_sampler *s)
{
    bool late = false;
    const hipError_t e = poll_stream(s->stream, &late);
    if (late) {  // (HIP_TRY reports the expression; the state goes to stderr once, here)
        s->wedged = true;
        if (s->pair) {
            std::lock_guard<std::mutex> g(g_pool_mu);
            s->pair->dead = true;
        }
        std::fprintf(stderr, "[occ] a copy or fill on the main stream did not complete within %.0f s: %s\n", host_wait_limit_s(), host_state(s).c_str());
    }
    return e;
}
static hipError_t copy_on(occ_sampler *s, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (!s->stream) return hipMemcpy(dst, src, bytes, kind);
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s->stream);
    return e != hipSuccess ? e : drain_for_copy(s);
}
static hipError_t fill_on(occ_sampler *s, void *dst, int value, size_t bytes)
{
    if (!s->stream) return hipMemset(dst, value, bytes);
    const hipError_t e = hipMemsetAsync(dst, value, bytes, s->stream);
    return e != hipSuccess ? e : drain_for_copy(s);
}

template <class T>
int dev_alloc(occ_sampler *s, T **out, size_t count, bool zero = true)
{
    void *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (zero) HIP_TRY(fill_on(s, p, 0, bytes));
    *out = (T *)p;
    return OCC_OK;
}

template <class T>
int upload(occ_sampler *s, const T **out, const std::vector<T> &h, const char *name = "array")
{
    T *d = nullptr;
    int rc = dev_alloc(s, &d, h.size(), false);
    if (rc) return rc;
    if (!h.empty() && !s->defer_fixed) HIP_TRY(copy_on(s, d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    if (!h.empty()) {
        s->fixed_list.emplace_back((void *)d, h.size() * sizeof(T));
        s->fixed_names.push_back(name);
    }
    *out = d;
    return OCC_OK;
}

// caller pointer (host or device) -> host vector
template <class T>
int fetch(occ_sampler *s, std::vector<T> &h, const T *src, size_t count)
{
    h.resize(count);
    if (count == 0) return OCC_OK;
    if (src == nullptr) {
        s->err = "null input pointer";
        return OCC_E_BADARG;
    }
    HIP_TRY(copy_on(s, h.data(), src, count * sizeof(T), hipMemcpyDefault));
    return OCC_OK;
}

int set_error(occ_sampler *s, int code, const char *msg)
{
    s->err = msg;
    return code;
}

using KernelE = void (*)(const Ctx *, ChainScalars *, Slot *, int, int);
using KernelEI = void (*)(const Ctx *, ChainScalars *, Slot *, int, int, int);

KernelEI pick_beta_partial(int p)
{
    switch (p) {
        case 0: return k_beta_partial<0>;  // generic path (run-time p)
        case 1: return k_beta_partial<1>;
        case 2: return k_beta_partial<2>;
        case 3: return k_beta_partial<3>;
        case 4: return k_beta_partial<4>;
        case 5: return k_beta_partial<5>;
        case 6: return k_beta_partial<6>;
        case 7: return k_beta_partial<7>;
        default: return k_beta_partial<8>;
    }
}
#define OCC_PICK_P(NAME, p) \
    ((p) == 0 ? NAME<0> : (p) == 1 ? NAME<1> : (p) == 2 ? NAME<2> : (p) == 3 ? NAME<3> : (p) == 4 ? NAME<4> : (p) == 5 ? NAME<5> : (p) == 6 ? NAME<6> : (p) == 7 ? NAME<7> : NAME<8>)
KernelE pick_omega_b(int p) { return OCC_PICK_P(k_omega_b, p); }
using KernelRsrE = void (*)(const RsrArgs, const Ctx *, ChainScalars *, Slot *, int, int);
KernelRsrE pick_rsr_eta_beta(int p)
{
    return p == 1 ? k_rsr_eta_beta<1> : p == 2 ? k_rsr_eta_beta<2> : p == 3 ? k_rsr_eta_beta<3> : p == 4 ? k_rsr_eta_beta<4> : p == 5 ? k_rsr_eta_beta<5> : p == 6 ? k_rsr_eta_beta<6> : p == 7 ? k_rsr_eta_beta<7> : k_rsr_eta_beta<8>;
}
using KernelRsr = void (*)(const RsrArgs, int);
KernelRsr pick_rsr_solve(int m)
{
    switch ((m + 15) / 16) {
        case 1: return k_rsr_solve<1>;
        case 2: return k_rsr_solve<2>;
        case 3: return k_rsr_solve<3>;
        case 4: return k_rsr_solve<4>;
        case 5: return k_rsr_solve<5>;
        case 6: return k_rsr_solve<6>;
        case 7: return k_rsr_solve<7>;
        default: return k_rsr_solve<8>;
    }
}
KernelEI pick_z_ob(int p) { return OCC_PICK_P(k_z_ob, p); }
KernelEI pick_omega_a(int q)
{
    switch (q) {
        case 0: return k_omega_a<0>;  // generic path (run-time q)
        case 1: return k_omega_a<1>;
        case 2: return k_omega_a<2>;
        case 3: return k_omega_a<3>;
        case 4: return k_omega_a<4>;
        case 5: return k_omega_a<5>;
        case 6: return k_omega_a<6>;
        case 7: return k_omega_a<7>;
        default: return k_omega_a<8>;
    }
}

// ---- single launches (all chains) ----------------------------------------------------------------
#define OCC_ARGS s->ctx_dev, s->ctx.sc, s->ctx.slots, 0, e
static const char *kind_name(int kind)
{
    static const char *names[] = {"k_omega_b", "k_noise", "k_eta_init", "k_minres", "k_beta_partial", "k_omega_a", "k_alpha_draw",
                                  "k_z_ob", "k_iter", "k_gate", "k_rsr_gram", "k_rsr_solve", "k_rsr_eta_beta"};
    return (kind >= 0 && kind < (int)(sizeof(names) / sizeof(names[0]))) ? names[kind] : "kernel";
}

// One kernel launch (all chains).  A launch the runtime rejects (bad grid, too much dynamic LDS, ...) is reported at
// once -- also during stream capture -- instead of surfacing later as stale results or a barrier time-out.
// Grid of a kernel that hands out its tiles in proportion to the CUs its stream owns on each XCD (tile_of_block_shared,
// kernel `kid` of Ctx::tile_first): 8 x the largest share; the plain (per_chain, C) grid when the XCDs are treated alike.
static dim3 shared_grid(const Ctx &c, int kid, int per_chain)
{
    if (!c.share_on) return dim3((unsigned)per_chain, (unsigned)c.C);
    return dim3(8u * (unsigned)c.tile_most[kid]);
}

int launch_kind(occ_sampler *s, hipStream_t st, int kind, int e, int extra = 0)
{
    const Ctx &c = s->ctx;
    const dim3 blk((unsigned)s->tpb), gs((unsigned)c.nb_n, (unsigned)c.C), gr((unsigned)c.nb_r, (unsigned)c.C);
    const int tp = s->generic ? 0 : c.p, tq = s->generic ? 0 : c.q;  // template arguments: 0 = the generic (run-time) instantiation
    const size_t lds_p = s->generic ? generic_lds_bytes(nacc(c.p), s->tpb) : 0, lds_q = s->generic ? generic_lds_bytes(nacc(c.q), s->tpb) : 0;
    switch (kind) {
        case K_OMEGA_B: hipLaunchKernelGGL(pick_omega_b(tp), gs, blk, 0, st, OCC_ARGS); break;
        case K_NOISE:
            // (256-thread blocks whatever the other kernels take: the first block of a chain also draws alpha, a wave per quantity)
            hipLaunchKernelGGL(k_noise, shared_grid(c, 2, (c.n + 255) / 256), dim3(256), 0, st, OCC_ARGS, extra, (extra == 1 && s->launch_sync) ? 1 : 0);
            if (c.dense_F != nullptr) {  // reference-form prior draw: uprior = F eps2, four chains per pass over F
                const dim3 gd((unsigned)((c.n + 3) / 4));
                for (int ch0 = 0; ch0 < c.C; ch0 += 4) hipLaunchKernelGGL(k_prior_dense<4>, gd, dim3(256), 0, st, s->ctx_dev, s->ctx.sc, ch0, e, extra);
            }
            break;
        case K_ETA_INIT: hipLaunchKernelGGL(k_eta_init<0>, gs, blk, 0, st, OCC_ARGS); break;
        case K_MINRES: hipLaunchKernelGGL(k_minres<0>, gs, blk, 0, st, s->kry, 0, e, extra); break;
        case K_BETA_PARTIAL: hipLaunchKernelGGL(pick_beta_partial(tp), gs, blk, lds_p, st, OCC_ARGS, extra); break;
        case K_OMEGA_A: hipLaunchKernelGGL(pick_omega_a(tq), shared_grid(c, 1, c.nb_r), blk, lds_q, st, OCC_ARGS, extra); break;  // extra = 1: k_gate's work first
        case K_ALPHA_DRAW: hipLaunchKernelGGL(k_alpha_draw<0>, dim3((unsigned)c.C), dim3(512), 0, st, OCC_ARGS, s->launch_sync ? 1 : 0); break;
        case K_GATE: hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, s->ctx_dev, s->ctx.sc); break;
        case K_RSR_GRAM:
            if (s->rsr.m > RSR_MAX_DIM && !std::getenv("OCC_NO_GRAM32")) {  // large bases: 32 x 32 blocks of G, then the K'u workgroups alone
                // (8 waves per workgroup whatever the chain count: a chain's bits do not depend on how many chains run beside it)
                if (c.C > 1 && !std::getenv("OCC_GRAM32_ONE_CHAIN"))  // two chains per workgroup: K streamed once for both
                    hipLaunchKernelGGL((k_rsr_gram32<2, GRAM32_WAVES>), dim3((unsigned)rsr_gram32_blocks(s->rsr.m), (unsigned)((c.C + 1) / 2)), dim3(64 * GRAM32_WAVES), rsr_gram32_lds(GRAM32_WAVES), st, s->rsr, e, s->launch_sync ? 1 : 0);
                else
                    hipLaunchKernelGGL((k_rsr_gram32<1, GRAM32_WAVES>), dim3((unsigned)rsr_gram32_blocks(s->rsr.m), (unsigned)c.C), dim3(64 * GRAM32_WAVES), rsr_gram32_lds(GRAM32_WAVES), st, s->rsr, e, s->launch_sync ? 1 : 0);
                hipLaunchKernelGGL(k_rsrb_u, dim3((unsigned)((s->rsr.n + 255) / 256), (unsigned)c.C), dim3(256), 0, st, s->rsr, e, s->launch_sync ? 1 : 0);
                hipLaunchKernelGGL(k_rsrb_ktu, dim3((unsigned)s->rsr.m, (unsigned)((c.C + 3) / 4)), dim3(256), 0, st, s->rsr, e);
            } else {
                hipLaunchKernelGGL(k_rsr_gram, dim3((unsigned)rsr_gram_tiles(s->rsr.m), (unsigned)c.C), dim3(64 * GRAM_WAVES), 0, st, s->rsr, e, s->launch_sync ? 1 : 0);
            }
            break;
        case K_RSR_SOLVE:
            if (s->rsr.m <= RSR_MAX_DIM) {
                hipLaunchKernelGGL(pick_rsr_solve(s->rsr.m), dim3(1, (unsigned)c.C), dim3(256), sizeof(double) * rsr_solve_lds_doubles(s->rsr.m), st, s->rsr, e);
            } else {  // the m x m system in global memory, factorised panel by panel (occ_rsr.hpp, k_rsrb_*)
                const int m = s->rsr.m;
                hipLaunchKernelGGL(k_rsrb_tau, dim3((unsigned)((m + RSRB_QROWS - 1) / RSRB_QROWS), (unsigned)c.C), dim3(1024), 0, st, s->rsr, e);
                hipLaunchKernelGGL(k_rsrb_assemble, dim3((unsigned)m, (unsigned)c.C), dim3(256), 0, st, s->rsr, e);
                // the head (the first panel's factor and block row), then per panel step one launch: the step's trailing update
                // and the next step's panel work
                hipLaunchKernelGGL(k_rsrb_step, dim3((unsigned)((m + 31) / 32), 2u * (unsigned)c.C), dim3(256), 0, st, s->rsr, e, -1, c.C);
                for (int k0 = 0; k0 + RSR_PANEL < m; k0 += RSR_PANEL) {
                    const unsigned tt = (unsigned)((m - k0 - RSR_PANEL + 31) / 32);
                    hipLaunchKernelGGL(k_rsrb_step, dim3(tt, (tt + 1) * (unsigned)c.C), dim3(256), 0, st, s->rsr, e, k0, c.C);
                }
                hipLaunchKernelGGL(k_rsrb_solve, dim3(1, (unsigned)c.C), dim3(1024), 0, st, s->rsr, e);
            }
            break;
        case K_RSR_ETA_BETA: hipLaunchKernelGGL(pick_rsr_eta_beta(c.p), gs, blk, 0, st, s->rsr, OCC_ARGS); break;
        case K_ITER:
            if (s->tiles) {  // k_tiles: eight bands of B + 1 workgroups per chain (the surplus one of a band returns at once)
                const dim3 gt(XL_SLOTS * (unsigned)(s->tiles_B + 1), (unsigned)c.C), bt(TILE);
                const size_t lds = tiles_lds_bytes(s->tiles_T);
                const int fl = (s->launch_sync ? 1 : 0) | s->iter_flags_extra;
                const bool dia = s->kry.dia_n > 0;
                void (*kt)(const IterArgs, int, int) = s->tiles_T == 1   ? (dia ? k_tiles<8, 1, 1> : k_tiles<8, 1, 0>)
                                                       : s->tiles_T == 2 ? (dia ? k_tiles<8, 2, 1> : k_tiles<8, 2, 0>)
                                                       : s->tiles_T == 3 ? (dia ? k_tiles<8, 3, 1> : k_tiles<8, 3, 0>)
                                                                         : (dia ? k_tiles<8, 4, 1> : k_tiles<8, 4, 0>);
                if (dia && s->tiles_T == 4) {  // gathers of several tiles in flight together (occ_tiles.hpp, GB); OCC_TILES_GB: developer knob
                    const char *gb = std::getenv("OCC_TILES_GB");
                    const int g = gb ? std::atoi(gb) : TILES_GB_DEFAULT;
                    if (g == 2) kt = k_tiles<8, 4, 1, 2>;
                    else if (g == 4) kt = k_tiles<8, 4, 1, 4>;
                }
                if (s->ext_ev0) hipExtLaunchKernelGGL(kt, gt, bt, lds, st, s->ext_ev0, s->ext_ev1, 0, s->iter, e, fl);
                else hipLaunchKernelGGL(kt, gt, bt, lds, st, s->iter, e, fl);
            }
            else if (s->xcd_local) {  // eight chains (one per XCD) per launch; more chains: the next eight right behind
                for (int base = 0; base < c.C; base += XL_SLOTS) {
                    IterArgs ia = s->iter;
                    ia.chain_base = base;
                    if (s->ext_ev0 && base == 0) {  // occ_profile: the dispatch's own begin / end timestamps
                        const int fl = (s->launch_sync ? 1 : 0) | s->iter_flags_extra;
                        if (s->xl_wide == 1) hipExtLaunchKernelGGL((k_iter<8, 1, 1>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG_XL), 0, st, s->ext_ev0, s->ext_ev1, 0, ia, e, fl);
                        else if (s->xl_wide == 2) hipExtLaunchKernelGGL((k_iter<8, 1, 2>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG_XL), 0, st, s->ext_ev0, s->ext_ev1, 0, ia, e, fl);
                        else if (s->iter_window == 16) hipExtLaunchKernelGGL((k_iter<16, 1, 0>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG), 0, st, s->ext_ev0, s->ext_ev1, 0, ia, e, fl);
                        else hipExtLaunchKernelGGL((k_iter<8, 1, 0>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG), 0, st, s->ext_ev0, s->ext_ev1, 0, ia, e, fl);
                        continue;
                    }
                    if (s->xl_wide == 1) hipLaunchKernelGGL((k_iter<8, 1, 1>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG_XL), 0, st, ia, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
                    else if (s->xl_wide == 2) hipLaunchKernelGGL((k_iter<8, 1, 2>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG_XL), 0, st, ia, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
                    else if (s->iter_window == 16) hipLaunchKernelGGL((k_iter<16, 1, 0>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG), 0, st, ia, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
                    else hipLaunchKernelGGL((k_iter<8, 1, 0>), dim3(XL_SLOTS * (unsigned)(ia.nbg + 1)), dim3(ITER_WG), 0, st, ia, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
                }
            }
            else if (s->ext_ev0) {
                const int fl = (s->launch_sync ? 1 : 0) | s->iter_flags_extra;
                if (s->iter_window == 8) hipExtLaunchKernelGGL((k_iter<8, 0, 0>), dim3((unsigned)s->iter.nbg, (unsigned)c.C), dim3(ITER_WG), 0, st, s->ext_ev0, s->ext_ev1, 0, s->iter, e, fl);
                else hipExtLaunchKernelGGL((k_iter<16, 0, 0>), dim3((unsigned)s->iter.nbg, (unsigned)c.C), dim3(ITER_WG), 0, st, s->ext_ev0, s->ext_ev1, 0, s->iter, e, fl);
            }
            else if (s->iter_window == 8) hipLaunchKernelGGL((k_iter<8, 0, 0>), dim3((unsigned)s->iter.nbg, (unsigned)c.C), dim3(ITER_WG), 0, st, s->iter, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
            else hipLaunchKernelGGL((k_iter<16, 0, 0>), dim3((unsigned)s->iter.nbg, (unsigned)c.C), dim3(ITER_WG), 0, st, s->iter, e, (s->launch_sync ? 1 : 0) | s->iter_flags_extra);
            break;
        default:
            if (s->tpb == 64) {  // 64-site slices (fused paths): 256-thread blocks, beta once per block, partial sums still per slice
                const unsigned nb4 = (unsigned)((c.n + 255) / 256);
                if (s->generic) hipLaunchKernelGGL((k_beta_draw<0, 0>), dim3((unsigned)c.C), dim3(256), 0, st, OCC_ARGS);
                hipLaunchKernelGGL(pick_z_ob(tp), s->generic ? dim3(nb4 * 2, (unsigned)c.C) : shared_grid(c, 0, (int)nb4 * 2), dim3(256), 0, st, OCC_ARGS,
                                   (s->launch_sync ? 1 : 0) | 2 | s->zob_debug);
            } else {
                if (s->generic) hipLaunchKernelGGL((k_beta_draw<0, 0>), dim3((unsigned)c.C), dim3(256), 0, st, OCC_ARGS);
                else if (s->beta_split) hipLaunchKernelGGL(OCC_PICK_P(k_beta_draw, c.p), dim3((unsigned)c.C), dim3(64), 0, st, OCC_ARGS);
                hipLaunchKernelGGL(pick_z_ob(tp), dim3((unsigned)c.nb_n * 2, (unsigned)c.C), blk, 0, st, OCC_ARGS,
                                   (s->launch_sync ? 1 : 0) | s->zob_debug | (s->beta_split ? 4 : 0));
            }
            break;
    }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        s->err = std::string("launch of ") + kind_name(kind) + " failed: " + hipGetErrorString(le);
        return OCC_E_HIP;
    }
    return OCC_OK;
}
// Launches inside sequences (and stream captures, which an early return would leave open) record the first failure in
// s->launch_rc; the caller collects it at the end of the sequence / after hipStreamEndCapture with take_launch_rc().
#define LAUNCH(...)                                          \
    do {                                                     \
        const int lrc_ = launch_kind(__VA_ARGS__);           \
        if (lrc_ != OCC_OK && s->launch_rc == OCC_OK) {      \
            s->launch_rc = lrc_;                             \
            s->launch_err = s->err;                          \
        }                                                    \
    } while (0)
int take_launch_rc(occ_sampler *s)
{
    const int rc = s->launch_rc;
    if (rc != OCC_OK) s->err = s->launch_err;
    s->launch_rc = OCC_OK;
    return rc;
}

int read_scalars(occ_sampler *s, std::vector<ChainScalars> &h)
{
    h.resize(s->ctx.C);
    if (s->flag_sync) WAIT_TRY(s->side);  // its last k_noise is not waited for by the main stream
    if (!s->pin_sc) HIP_TRY(hipHostMalloc((void **)&s->pin_sc, 2 * sizeof(ChainScalars) * h.size(), hipHostMallocDefault));
    ChainScalars *back = s->pin_sc + h.size();
    HIP_TRY(hipMemcpyAsync(back, s->ctx.sc, sizeof(ChainScalars) * h.size(), hipMemcpyDeviceToHost, s->stream));
    WAIT_TRY(s->stream);
    std::memcpy(h.data(), back, sizeof(ChainScalars) * h.size());
    return OCC_OK;
}
int write_scalars(occ_sampler *s, const std::vector<ChainScalars> &h)
{
    s->mirror_ok = false;  // (whoever writes them has changed what the host's copy stood for)
    HIP_TRY(hipMemcpyAsync(s->ctx.sc, h.data(), sizeof(ChainScalars) * h.size(), hipMemcpyHostToDevice, s->stream));
    WAIT_TRY(s->stream);
    return OCC_OK;
}

int check_device_errors(occ_sampler *s, const std::vector<ChainScalars> &h)
{
    for (size_t c = 0; c < h.size(); ++c) {
        if (h[c].err == OCC_E_MINRES) { s->err = "MINRES solver did not converge!"; return OCC_E_MINRES; }
        if (h[c].err == OCC_E_CHOLESKY) { s->err = "Cholesky factorization/solver failed!"; return OCC_E_CHOLESKY; }
        if (h[c].err == OCC_E_HIP) {
            s->device_timeout = true;
            s->err = "a device-side wait timed out (a barrier among the workgroups of a chain in k_iter, or a hand-over "
                     "between the two streams): the device is over-subscribed or its queues are being serialised; "
                     "OCC_EVENT_SYNC=1 hands over through events, OCC_NO_PERSISTENT=1 uses one launch per MINRES step";
            return OCC_E_HIP;
        }
    }
    return OCC_OK;
}

// omega_b and the right-hand-side noise of the CURRENT iteration, stand-alone (after new start values).
int launch_prologue(occ_sampler *s)
{
    LAUNCH(s, s->stream, K_OMEGA_B, s->parity);
    LAUNCH(s, s->stream, K_NOISE, s->parity, 0);  // ahead = 0: outside the sequence counting
    s->need_prologue = false;
    return take_launch_rc(s);
}

// Krylov launches from `k_from` with the host watching the `done` flags.  On return *k_last is the
// last launch (every chain's final scalars live in slot k_last & 3).
int eager_krylov(occ_sampler *s, int k_from, int *k_last)
{
    std::vector<Slot> slots((size_t)s->ctx.C * NSLOT);
    for (int k = k_from;; ++k) {
        LAUNCH(s, s->stream, K_MINRES, s->parity, k);
        if (k < 4) continue;
        HIP_TRY(hipMemcpyAsync(slots.data(), s->ctx.slots, sizeof(Slot) * slots.size(), hipMemcpyDeviceToHost, s->stream));
        WAIT_TRY(s->stream);
        if (int lrc = take_launch_rc(s)) return lrc;
        bool all = true;
        for (int c = 0; c < s->ctx.C; ++c) all = all && slots[(size_t)c * NSLOT + (k & (NSLOT - 1))].done;
        if (all) {
            *k_last = k;
            return OCC_OK;
        }
        if ((long long)k > s->ctx.maxiter + 3) return set_error(s, OCC_E_MINRES, "MINRES solver did not converge!");
    }
}

// One launch sequence on the main stream only, every kernel in a valid topological order of the DAG
// (the reference's own order of conditionals, logit.py:254-266, with omega_a/alpha moved up front --
// their inputs are last iteration's alpha and z).
// One iteration of the reduced-rank model on one stream, the reference's order (omega_a / alpha moved up front as
// everywhere; their inputs are last iteration's alpha and z).
int launch_rsr_sequence(occ_sampler *s, hipStream_t st, int e)
{
    LAUNCH(s, st, K_OMEGA_A, e);
    LAUNCH(s, st, K_NOISE, e, 1);
    LAUNCH(s, st, K_RSR_GRAM, e);
    LAUNCH(s, st, K_RSR_SOLVE, e);
    LAUNCH(s, st, K_RSR_ETA_BETA, e);
    LAUNCH(s, st, K_Z_OB, e);
    return OCC_OK;
}

int eager_sequence(occ_sampler *s)
{
    if (s->need_prologue) {
        const int prc = launch_prologue(s);
        if (prc) return prc;
    }
    const int e = s->parity;
    // one stream, reference order: stream order is the synchronisation, the hand-over counters stay untouched
    struct NoSync { occ_sampler *s; bool old; explicit NoSync(occ_sampler *p) : s(p), old(p->launch_sync) { s->launch_sync = false; } ~NoSync() { s->launch_sync = old; } } no_sync(s);
    if (s->rsr.m > 0) {  // reduced-rank model: the theta conditional stands where the ICAR solve is
        launch_rsr_sequence(s, s->stream, e);
        s->parity ^= 1;
        s->eager_iterations += 1;
        return take_launch_rc(s);
    }
    LAUNCH(s, s->stream, K_OMEGA_A, e);
    LAUNCH(s, s->stream, K_NOISE, e, 1);
    if (s->persistent) {
        LAUNCH(s, s->stream, K_ITER, e);
    } else {
        LAUNCH(s, s->stream, K_ETA_INIT, e);
        int k_last = 0;
        int rc = eager_krylov(s, 1, &k_last);
        if (rc) return rc;
        s->calib_max = std::max(s->calib_max, k_last - 3);
        LAUNCH(s, s->stream, K_BETA_PARTIAL, e, k_last);
    }
    LAUNCH(s, s->stream, K_Z_OB, e);
    s->parity ^= 1;
    s->eager_iterations += 1;
    return take_launch_rc(s);
}

void destroy_head(occ_sampler *s)
{
    for (int e = 0; e < 2; ++e) {
        if (s->head[e]) (void)hipGraphExecDestroy(s->head[e]);
        if (s->head_graph[e]) (void)hipGraphDestroy(s->head_graph[e]);
        s->head[e] = nullptr;
        s->head_graph[e] = nullptr;
    }
}

void destroy_graph(occ_sampler *s)
{
    destroy_head(s);
    for (int e = 0; e < 2; ++e) {
        if (s->tail[e]) (void)hipGraphExecDestroy(s->tail[e]);
        if (s->tail_graph[e]) (void)hipGraphDestroy(s->tail_graph[e]);
        s->tail[e] = nullptr;
        s->tail_graph[e] = nullptr;
    }
}

// The last node of a linear graph (the only node nothing depends on).
int graph_leaf(occ_sampler *s, hipGraph_t graph, hipGraphNode_t *leaf)
{
    size_t n = 0;
    HIP_TRY(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    HIP_TRY(hipGraphGetNodes(graph, nodes.data(), &n));
    for (hipGraphNode_t nd : nodes) {
        size_t deps = 0;
        HIP_TRY(hipGraphNodeGetDependentNodes(nd, nullptr, &deps));
        if (deps == 0) { *leaf = nd; return OCC_OK; }
    }
    return set_error(s, OCC_E_HIP, "captured graph has no leaf node");
}

// Capture the per-parity chains.  Iteration j of a solve is tested by launch j + 3, so `cap` iterations
// need cap + 3 Krylov launches; a solve that needs more is carried into the next sequence by the kernels
// themselves (Ctl::koff), so `cap` trades empty launches against carried sequences.  Only the head
// depends on `cap`; the side chains are captured once.
int build_graph(occ_sampler *s, int cap)
{
    const bool verbose = std::getenv("OCC_VERBOSE") != nullptr;
    const auto host_t0 = std::chrono::steady_clock::now();
    WAIT_TRY(s->stream);
    destroy_head(s);
    int rc;
    if (s->rsr.m > 0 && !s->flag_sync) {  // reduced-rank model: GRAPH_SEQ iterations (alternating parity) on the main stream
        const bool old_sync = s->launch_sync;
        s->launch_sync = false;
        HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < GRAPH_SEQ; ++t) launch_rsr_sequence(s, s->stream, s->parity ^ (t & 1));
        HIP_TRY(hipStreamEndCapture(s->stream, &s->head_graph[0]));
        s->launch_sync = old_sync;
        HIP_TRY(hipGraphInstantiate(&s->head[0], s->head_graph[0], nullptr, nullptr, 0));
        s->graph_parity = s->parity;
        s->krylov_cap = 0;
        return take_launch_rc(s);
    }
    if (s->flag_sync) {
        // GRAPH_SEQ sequences (alternating parity) per graph and stream, no event nodes: the kernels hand over through
        // the device counters of Ctx::sync.  The counters restart with the capture: the main stream's sequence numbers
        // live in two words indexed by the sequence PARITY, and a capture that starts with the other parity than the last
        // one ended with (an odd number of stepped iterations in between: occ_step hands over by stream order and leaves
        // the counters alone) would read the older of the two -- one sequence behind the side stream, whose gate then
        // waits for a number the main stream never announces (found by the MINRES-limit test, round 3).  Both streams are
        // idle here.  (SYNC_DEBUG, the tests' broken-hand-over word, is not a counter and stays.)
        WAIT_TRY(s->side);
        HIP_TRY(fill_on(s, s->ctx.sync, 0, sizeof(unsigned) * SYNC_DEBUG));
        HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < GRAPH_SEQ; ++t) {
            const int e = s->parity ^ (t & 1);
            if (s->rsr.m > 0) {  // reduced-rank model: k_rsr_gram opens the sequence as k_iter does
                LAUNCH(s, s->stream, K_RSR_GRAM, e);
                LAUNCH(s, s->stream, K_RSR_SOLVE, e);
                LAUNCH(s, s->stream, K_RSR_ETA_BETA, e);
            } else {
                LAUNCH(s, s->stream, K_ITER, e);
            }
            LAUNCH(s, s->stream, K_Z_OB, e);
        }
        HIP_TRY(hipStreamEndCapture(s->stream, &s->head_graph[0]));
        HIP_TRY(hipGraphInstantiate(&s->head[0], s->head_graph[0], nullptr, nullptr, 0));
        HIP_TRY(hipStreamBeginCapture(s->side, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < GRAPH_SEQ; ++t) {
            if (std::getenv("OCC_GATE_KERNEL")) {  // diagnostic: the gate as a kernel of its own, as until round 3
                LAUNCH(s, s->side, K_GATE, 0);
                LAUNCH(s, s->side, K_OMEGA_A, s->parity ^ (t & 1));
            } else {
                LAUNCH(s, s->side, K_OMEGA_A, s->parity ^ (t & 1), 1);
            }
            LAUNCH(s, s->side, K_NOISE, s->parity ^ (t & 1), 1);
        }
        HIP_TRY(hipStreamEndCapture(s->side, &s->tail_graph[0]));
        HIP_TRY(hipGraphInstantiate(&s->tail[0], s->tail_graph[0], nullptr, nullptr, 0));
        s->graph_parity = s->parity;
        s->krylov_cap = 0;
        return take_launch_rc(s);
    }
    for (int e = 0; e < 2; ++e) {
        HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        if (s->persistent) {
            LAUNCH(s, s->stream, K_ITER, e);
        } else {
            LAUNCH(s, s->stream, K_ETA_INIT, e);
            for (int k = 1; k <= cap + 3; ++k) LAUNCH(s, s->stream, K_MINRES, e, k);
            LAUNCH(s, s->stream, K_BETA_PARTIAL, e, cap + 3);
        }
        HIP_TRY(hipStreamEndCapture(s->stream, &s->head_graph[e]));
        if (s->event_nodes) {  // ... -> wait(side chain of this iteration) -> k_z_ob -> record
            hipGraphNode_t leaf, wait, rec;
            if ((rc = graph_leaf(s, s->head_graph[e], &leaf))) return rc;
            HIP_TRY(hipGraphAddEventWaitNode(&wait, s->head_graph[e], &leaf, 1, s->ev_side[e]));
            HIP_TRY(hipStreamBeginCaptureToGraph(s->stream, s->head_graph[e], &wait, nullptr, 1, hipStreamCaptureModeThreadLocal));
            LAUNCH(s, s->stream, K_Z_OB, e);
            hipGraph_t same = nullptr;
            HIP_TRY(hipStreamEndCapture(s->stream, &same));
            if ((rc = graph_leaf(s, s->head_graph[e], &leaf))) return rc;
            HIP_TRY(hipGraphAddEventRecordNode(&rec, s->head_graph[e], &leaf, 1, s->ev_z[e]));
        }
        HIP_TRY(hipGraphInstantiate(&s->head[e], s->head_graph[e], nullptr, nullptr, 0));
        if (s->tail[e]) continue;
        if (s->event_nodes) {  // wait(previous k_z_ob) -> side chain -> record
            hipGraphNode_t leaf, wait, rec;
            HIP_TRY(hipGraphCreate(&s->tail_graph[e], 0));
            HIP_TRY(hipGraphAddEventWaitNode(&wait, s->tail_graph[e], nullptr, 0, s->ev_z[e ^ 1]));
            HIP_TRY(hipStreamBeginCaptureToGraph(s->side, s->tail_graph[e], &wait, nullptr, 1, hipStreamCaptureModeThreadLocal));
            LAUNCH(s, s->side, K_OMEGA_A, e);
            LAUNCH(s, s->side, K_NOISE, e, 1);
            hipGraph_t same = nullptr;
            HIP_TRY(hipStreamEndCapture(s->side, &same));
            if ((rc = graph_leaf(s, s->tail_graph[e], &leaf))) return rc;
            HIP_TRY(hipGraphAddEventRecordNode(&rec, s->tail_graph[e], &leaf, 1, s->ev_side[e]));
        } else {
            HIP_TRY(hipStreamBeginCapture(s->side, hipStreamCaptureModeThreadLocal));
            LAUNCH(s, s->side, K_OMEGA_A, e);
            LAUNCH(s, s->side, K_NOISE, e, 1);
            HIP_TRY(hipStreamEndCapture(s->side, &s->tail_graph[e]));
        }
        HIP_TRY(hipGraphInstantiate(&s->tail[e], s->tail_graph[e], nullptr, nullptr, 0));
    }
    s->krylov_cap = cap;
    if (verbose) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
        std::fprintf(stderr, "[occ] graphs captured, Krylov cap %d: %.2f ms host time\n", cap, ms);
    }
    return take_launch_rc(s);
}

// Enqueue one launch sequence (one Gibbs iteration of every chain, or a carried solve) without any
// host synchronisation.  DAG: side chain after the previous k_z_ob; k_z_ob after head and side chain.
int enqueue_sequence(occ_sampler *s)
{
    const int e = s->parity;
    if (s->flag_sync) {  // GRAPH_SEQ sequences on each stream
        HIP_TRY(hipGraphLaunch(s->tail[0], s->side));
        HIP_TRY(hipGraphLaunch(s->head[0], s->stream));
        return OCC_OK;
    }
    if (s->rsr.m > 0) {  // GRAPH_SEQ sequences
        HIP_TRY(hipGraphLaunch(s->head[0], s->stream));
        return OCC_OK;
    }
    if (s->event_nodes) {  // the graphs carry their own waits and records
        HIP_TRY(hipGraphLaunch(s->tail[e], s->side));
        HIP_TRY(hipGraphLaunch(s->head[e], s->stream));
        s->parity ^= 1;
        return OCC_OK;
    }
    if (s->side_enabled) {
        HIP_TRY(hipStreamWaitEvent(s->side, s->ev_z[e ^ 1], 0));
        HIP_TRY(hipGraphLaunch(s->tail[e], s->side));
        HIP_TRY(hipEventRecord(s->ev_side[e], s->side));
        HIP_TRY(hipGraphLaunch(s->head[e], s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_side[e], 0));
    } else {  // diagnostic (OCC_NO_SIDE_STREAM): the same kernels in one stream
        HIP_TRY(hipGraphLaunch(s->tail[e], s->stream));
        HIP_TRY(hipGraphLaunch(s->head[e], s->stream));
    }
    LAUNCH(s, s->stream, K_Z_OB, e);
    if (s->side_enabled) HIP_TRY(hipEventRecord(s->ev_z[e], s->stream));
    s->parity ^= 1;
    return take_launch_rc(s);
}

// The chains' scalars: the host's mirror when it still stands for the device (mirror_ok), else a read.
int get_scalars(occ_sampler *s, std::vector<ChainScalars> &h)
{
    if (s->mirror_ok) { h = s->mirror; return OCC_OK; }
    return read_scalars(s, h);
}

// Head of every occ_run / occ_step / occ_profile: opens the call's WINDOW of iterations in the chains' scalars on the
// device -- it_base = the iteration the chain is at, it_stop = it_base + n_iter, burnin, keep, no carried solve -- in
// front of the call's kernels on the main stream, without waiting for it.  Two forms, one decision (round 3 spread it
// over snapshot_take / set_window and five host flags):
//   snapshot (paths with device-side waits: the call can be re-run): ONE kernel, k_snapshot, copies eta, z, x (theta) of every
//     chain to the snapshot buffers, resets k_iter's clock words (run = true) and makes the window's edits ON THE DEVICE;
//     the scalars as they were BEFORE the edits stay on the host in snap_sc (what fallback_to_launch_per_step restores);
//   upload (everything else): the host edits its copy of the scalars and sends it (page-locked staging, asynchronous).
// The host's mirror is dropped either way: the kernels that follow change the scalars.
int open_window(occ_sampler *s, int64_t n_iter, int64_t burnin, int64_t keep, bool snapshot, bool run)
{
    const Ctx &c = s->ctx;
    std::vector<ChainScalars> h;
    int rc = get_scalars(s, h);
    if (rc) return rc;
    s->mirror_ok = false;
    // (the slot counters of the one-XCD forms are zero between sequences -- k_z_ob resets them; a call that ended in an
    // error may have left them anywhere)
    if (!s->clean_exit) {  // (the last call through here did not end cleanly, or there was none)
        if (c.claim) HIP_TRY(hipMemsetAsync(c.claim, 0, sizeof(unsigned) * (size_t)c.C * 16, s->stream));
        if (c.sync) HIP_TRY(hipMemsetAsync(c.sync + SYNC_ABORT, 0, sizeof(unsigned), s->stream));
    }
    s->clean_exit = false;  // until the call says otherwise
    // tests (occ_set_state "debug_close_window"): the device gets a window of ZERO iterations while the host expects n_iter
    const uint32_t dev_n = s->debug_close_window ? 0u : (uint32_t)n_iter;
    s->debug_close_window = false;
    if (snapshot) {
        const size_t Cn = (size_t)c.C * c.n;
        if (!s->snap_eta) {
            if ((rc = dev_alloc(s, &s->snap_eta, Cn, false))) return rc;
            if ((rc = dev_alloc(s, &s->snap_z, Cn, false))) return rc;
            if ((rc = dev_alloc(s, &s->snap_x, Cn, false))) return rc;
            if (s->rsr.m > 0 && (rc = dev_alloc(s, &s->snap_theta, (size_t)c.C * s->rsr.m, false))) return rc;
        }
        s->snap_sc = h;
        s->snap_parity = s->parity;
        hipLaunchKernelGGL(k_snapshot, dim3((unsigned)std::min<size_t>((Cn + 255) / 256, 2048)), dim3(256), 0, s->stream, c.eta, s->snap_eta, c.z, s->snap_z, c.Xv, s->snap_x,
                           (unsigned long long)Cn, s->rsr.m > 0 ? s->rsr.theta : nullptr, s->snap_theta, (unsigned long long)c.C * (unsigned long long)std::max(s->rsr.m, 0),
                           run ? c.iter_clock : nullptr, c.sc, c.C, s->parity, dev_n, (uint32_t)burnin, (uint32_t)keep);
        if (hipGetLastError() != hipSuccess) return set_error(s, OCC_E_HIP, "launch of k_snapshot failed");
        return OCC_OK;
    }
    for (auto &sc : h) {
        Ctl &ctl = sc.ctl[s->parity];
        sc.it_base = ctl.it;
        sc.it_stop = ctl.it + dev_n;
        sc.burnin = (uint32_t)burnin;
        sc.keep = (uint32_t)keep;
        ctl.koff = 0;
    }
    if (!s->pin_sc) HIP_TRY(hipHostMalloc((void **)&s->pin_sc, 2 * sizeof(ChainScalars) * h.size(), hipHostMallocDefault));
    std::memcpy(s->pin_sc, h.data(), sizeof(ChainScalars) * h.size());  // (free again: the last copy from it was followed by a wait for the stream)
    HIP_TRY(hipMemcpyAsync(c.sc, s->pin_sc, sizeof(ChainScalars) * h.size(), hipMemcpyHostToDevice, s->stream));
    if (run && c.iter_clock) {  // k_iter's clock counts this run only
        HIP_TRY(hipMemsetAsync(c.iter_clock, 0xff, sizeof(unsigned long long), s->stream));       // [0] = ~0: the earliest start so far
        HIP_TRY(hipMemsetAsync(c.iter_clock + 1, 0, 3 * sizeof(unsigned long long), s->stream));  // [1] latest end, [2] sum, [3] launches
    }
    return OCC_OK;
}


// ---- the pooled stream pairs (see StreamPair above) --------------------------------------------------------------------
// -> the pair with these masks on `device` (created when no engine holds one yet); nullptr when it cannot be had -- the cap
// on masked pairs is reached, or the runtime cannot create the streams -- with the reason in *why.  The device is current.
// At process exit: the idle pairs go before the runtime does (registered at the first acquisition, so it runs ahead of the
// finalisers of everything loaded before this library; a profiler that walks the live streams at its own finalisation --
// rocprofv3 7.2 does -- otherwise meets streams nobody will ever destroy: SIGSEGV inside __cxa_finalize).
static void pool_at_exit()
{
    std::lock_guard<std::mutex> g(g_pool_mu);
    for (auto &kv : g_slots) {
        for (StreamPair *q : kv.second->pairs) {
            // held pairs too (an engine some Python object still owns, an exception path of a tool): their streams must not
            // outlive this hook either.  The handles are nulled and the pair marked dead; an occ_destroy that comes later
            // (interpreter finalisation) finds no stream to wait for or destroy.
            if (q->dead || !q->main) { q->dead = true; continue; }
            (void)hipSetDevice(q->device);
            bool late_m = false, late_s = false;
            (void)poll_stream(q->main, &late_m);
            (void)poll_stream(q->side, &late_s);
            q->dead = true;
            if (late_m || late_s) continue;  // (streams that never drain are left to the runtime)
            (void)hipStreamDestroy(q->side);
            (void)hipStreamDestroy(q->main);
            q->main = q->side = nullptr;
        }
    }
}

StreamPair *acquire_pair(int device, const std::vector<uint32_t> &m_main, const std::vector<uint32_t> &m_side, std::string *why)
{
    DeviceSlot &slot = device_slot(device);
    static std::once_flag at_exit_once;
    std::call_once(at_exit_once, [] { std::atexit(pool_at_exit); });
    std::lock_guard<std::mutex> g(g_pool_mu);
    int masked = 0, cap = MAX_MASKED_PAIRS;
    if (const char *mc = std::getenv("OCC_MAX_MASKED_PAIRS")) cap = std::max(0, std::atoi(mc));  // tests
    int live_masked = 0;
    StreamPair *match = nullptr;
    for (StreamPair *p : slot.pairs) {
        if (p->dead) continue;  // (its queues are not being served, or the process is on its way out)
        if (p->m_main == m_main && p->m_side == m_side) match = p;
        masked += p->m_main.empty() ? 0 : 1;
        live_masked += (!p->m_main.empty() && p->refs > 0) ? 1 : 0;
    }
    // (an idle pair -- refs 0, kept by release_pair -- comes back into use as it is; the cap is on the partitions engines HOLD)
    if (match && (match->refs > 0 || m_main.empty() || live_masked < cap)) {
        match->refs += 1;
        return match;
    }
    // at the cap: idle masked pairs (no engine holds them) make room first
    if (!m_main.empty() && masked >= cap) {
        for (size_t i = 0; i < slot.pairs.size() && masked >= cap;) {
            StreamPair *q = slot.pairs[i];
            if (q->refs == 0 && !q->m_main.empty() && !q->dead) {
                // (idle: release_pair's caller drained both streams before it let go, and nothing has been enqueued since)
                slot.pairs.erase(slot.pairs.begin() + (long)i);
                (void)hipStreamDestroy(q->side);
                (void)hipStreamDestroy(q->main);
                delete q;
                g_stream_gen.fetch_add(1);
                g_pairs_evicted.fetch_add(1);
                masked -= 1;
            } else {
                ++i;
            }
        }
    }
    if (!m_main.empty() && masked >= cap) {
        *why = std::to_string(masked) + " CU-masked stream pairs are alive on device " + std::to_string(device) +
               " already (each masked stream holds one of the device's 24 hardware queues)";
        return nullptr;
    }
    StreamPair *p = new StreamPair();
    p->device = device;
    p->m_main = m_main;
    p->m_side = m_side;
    hipError_t e;
    if (m_main.empty()) {  // the main stream (critical path) at the higher priority
        int prio_low = 0, prio_high = 0;
        e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
        if (std::getenv("OCC_NO_STREAM_PRIORITY")) prio_high = prio_low;
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->main, hipStreamNonBlocking, prio_high);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_low);
    } else {
        e = hipExtStreamCreateWithCUMask(&p->main, (uint32_t)m_main.size(), m_main.data());
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&p->side, (uint32_t)m_side.size(), m_side.data());
    }
    if (e != hipSuccess) {  // (an error here is not sticky)
        *why = std::string("stream creation failed: ") + hipGetErrorString(e);
        if (p->main) (void)hipStreamDestroy(p->main);
        if (p->side) (void)hipStreamDestroy(p->side);
        (void)hipGetLastError();
        delete p;
        return nullptr;
    }
    p->refs = 1;
    slot.pairs.push_back(p);
    g_stream_gen.fetch_add(1);
    return p;
}

void release_pair(StreamPair *p)
{
    if (!p) return;
    DeviceSlot &slot = device_slot(p->device);
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (--p->refs > 0) return;
    // The last engine that held the pair is gone: the streams stay, idle, for the next engine that wants this partition (a
    // process that creates and closes engines one after the other -- a test-suite, a parameter sweep -- then keeps ONE set of
    // hardware queues instead of creating and destroying CU-masked streams by the dozen; the stops inside occ_create seen
    // in round 3, HISTORY.md, came from runs that did exactly that).  acquire_pair destroys idle pairs when it needs their
    // place under the cap; the rest go with the process.
    (void)slot;  // (drop_pair has drained both streams)
}

// live pairs of a device: {masked, unmasked} (occ_stats, OCC_VERBOSE)
void count_pairs(int device, int *masked, int *plain, int *idle = nullptr)
{
    DeviceSlot &slot = device_slot(device);
    std::lock_guard<std::mutex> g(g_pool_mu);
    *masked = *plain = 0;
    if (idle) *idle = 0;
    for (StreamPair *p : slot.pairs) {
        if (p->dead) continue;
        if (p->refs > 0) (p->m_main.empty() ? *plain : *masked) += 1;  // (idle pairs wait for a taker: not live)
        else if (idle) *idle += 1;
    }
}

void adopt_pair(occ_sampler *s, StreamPair *p)
{
    s->pair = p;
    s->stream = p ? p->main : nullptr;
    s->side = p ? p->side : nullptr;
}

// The engine lets go of its streams (everything it enqueued has completed).
void drop_pair(occ_sampler *s)
{
    if (s->pair && !s->pair->dead) {  // (a pair whose queues are not served stays out of the pool's hands: acquire_pair skips it)
        (void)wait_on(s, s->stream, __func__, __LINE__);
        (void)wait_on(s, s->side, __func__, __LINE__);
    }
    release_pair(s->pair);
    adopt_pair(s, nullptr);
}

// The two streams without a CU partition (the pooled unmasked pair of the device).
int create_plain_streams(occ_sampler *s)
{
    std::string why;
    StreamPair *p = acquire_pair(s->device, {}, {}, &why);
    if (!p) return set_error(s, OCC_E_HIP, why.c_str());
    adopt_pair(s, p);
    s->main_cus = 0;
    s->flag_sync = false;
    s->ctx.share_on = 0;
    return OCC_OK;
}

// ICAR model on the launch-per-step path: no CU partition, hand-overs by events (the device-counter hand-overs and
// the partition exist for the fused kernel's sake).
int demote_streams(occ_sampler *s, bool also_reduced_rank = false)
{
    if ((s->rsr.m > 0 && !also_reduced_rank) || s->main_cus == 0) return OCC_OK;
    drop_pair(s);
    return create_plain_streams(s);
}

// Do the engine's two streams run BESIDE each other?  The device-side hand-overs presume it.  A kernel on the side stream
// waits (at most ~20 ms) for a word that a kernel launched AFTER it on the main stream sets: streams that share a hardware
// queue, or whose queues the scheduler is time-slicing (more live queues on the device than hardware slots), show up as
// "never seen" or as a wait of a scheduling quantum.  Asked at creation and again whenever the process's set of
// streams has changed since the last answer (g_stream_gen).
int stream_probe(occ_sampler *s, bool *beside)
{
    *beside = true;
    s->probe_gen = g_stream_gen.load();
    s->stream_probes += 1;
    if (std::getenv("OCC_DEBUG_SKIP_STREAM_PROBE")) return OCC_OK;
    int rc;
    if (!s->probe_w && (rc = dev_alloc(s, &s->probe_w, 32))) return rc;
    unsigned seen[2] = {0u, 0u};
    // (the pair of launches is timed: first a launch on each stream that is not -- the kernels' code object is loaded at a
    // process's first launch, milliseconds, and a queue's very first packet costs 0.5 ms)
    hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(64), 0, s->side, s->probe_w + 24);
    hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(64), 0, s->stream, s->probe_w + 24);
    WAIT_TRY(s->side);
    HIP_TRY(hipMemsetAsync(s->probe_w, 0, 32 * sizeof(unsigned), s->stream));
    WAIT_TRY(s->stream);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_stream_probe_wait, dim3(1), dim3(64), 0, s->side, s->probe_w);
    hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(64), 0, s->stream, s->probe_w);
    WAIT_TRY(s->stream);
    WAIT_TRY(s->side);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(copy_on(s, seen, s->probe_w + 16, sizeof(seen), hipMemcpyDeviceToHost));
    // The verdict is the DEVICE's: how many polls (about 2 us each) the waiting kernel made before it saw the word.  Beside
    // each other: a few dozen (the second launch's way through its queue); a queue that has to be scheduled in first costs a
    // quantum -- thousands of polls (10 ms and more, tools/queue_probe.hip) -- or the word is never seen.  (Round 3 also
    // looked at the HOST's clock around the pair of launches: a pre-empted host thread or a profiler's interception then
    // read as "not beside each other", and stayed.)
    *beside = seen[0] != 0u && seen[1] < 1024u;
    if (std::getenv("OCC_DEBUG_STREAMS_SERIALISED")) *beside = false;  // tests: take the caller's "not beside" branch
    if (std::getenv("OCC_VERBOSE")) {
        int masked = 0, plain = 0;
        count_pairs(s->device, &masked, &plain);
        std::fprintf(stderr, "[occ] stream probe: word %s after %u polls, %.0f us; live stream pairs on device %d: %d CU-masked, %d unmasked -> %s\n",
                     seen[0] ? "seen" : "NOT seen", seen[1], us, s->device, masked, plain, *beside ? "beside each other" : "NOT beside each other");
    }
    return OCC_OK;
}

// CUs of the main stream with k_tiles: half the device (whole shader engines per XCD), OCC_TILES_MAIN_CUS overrides
int tiles_main_cus(int ncu)
{
    int m = (ncu / 64) * 32;
    if (const char *e = std::getenv("OCC_TILES_MAIN_CUS")) m = std::max(32, std::min((std::atoi(e) / 32) * 32, ncu - 32));
    return m;
}

// k_tiles' invariant between launches (canaries in record buffer 1): established at creation and
// after anything that may have left the buffers otherwise.
int tiles_reset(occ_sampler *s)
{
    const Ctx &c = s->ctx;
    hipLaunchKernelGGL(k_tiles_reset, dim3((unsigned)((c.n + 255) / 256), (unsigned)c.C), dim3(256), 0, s->stream, s->iter);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return set_error(s, OCC_E_HIP, "launch of k_tiles_reset failed");
    WAIT_TRY(s->stream);
    return OCC_OK;
}

// Residency probe of the fused iteration kernel in the form s->xcd_local / xl_wide / iter_window / iter.nbg select:
// k_iter itself (flags bit 1) -- same grid, registers and LDS as the real launch -- runs ONE barrier among the
// workgroups of every chain with a short time limit, three times.  It passes exactly when every chain's workgroups
// are resident together on the CUs the main stream owns (and, one XCD per chain, share that XCD: a flag carries its
// writer's XCC_ID).  The condition it checks: workgroups per chain <= (workgroups the kernel's registers and LDS allow
// per CU) x (CUs of the stream's mask that the dispatcher gives the chain) -- per XCD and per shader engine, which
// only the hardware knows (26 workgroups on a 26-CU-per-XCD mask dead-locked, on 28 they ran).
int residency_probe(occ_sampler *s, bool *ok)
{
    *ok = true;
    std::vector<ChainScalars> h;
    int rc;
    for (int rep = 0; rep < 3 && *ok; ++rep) {
        s->iter_flags_extra = 2;
        HIP_TRY(fill_on(s, s->ctx.claim, 0, sizeof(unsigned) * (size_t)s->ctx.C * 16));
        LAUNCH(s, s->stream, K_ITER, 0);
        s->iter_flags_extra = 0;
        if ((rc = take_launch_rc(s))) return rc;
        if ((rc = read_scalars(s, h))) return rc;
        for (auto &sc : h)
            if (sc.err != 0) *ok = false;
    }
    // the barrier state starts from scratch whatever the probes left behind
    HIP_TRY(fill_on(s, s->ctx.bar, 0, sizeof(unsigned) * (size_t)s->ctx.C * BAR_STRIDE));
    HIP_TRY(fill_on(s, s->ctx.claim, 0, sizeof(unsigned) * (size_t)s->ctx.C * 16));
    if ((rc = read_scalars(s, h))) return rc;
    for (auto &sc : h) { sc.bar_base = s->tiles ? sc.bar_base + 16u : 0u; sc.err = 0; }  // (k_tiles: the tag of its tagged records, never reused)
    if ((rc = write_scalars(s, h))) return rc;
    return s->tiles ? tiles_reset(s) : OCC_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int32_t occ_abi_version(void) { return OCC_ABI_VERSION; }

int32_t occ_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *occ_last_error(const occ_sampler *s) { return s ? s->err.c_str() : g_create_error.c_str(); }

int occ_destroy(occ_sampler *s)
{
    if (!s) return OCC_OK;
    {
        DeviceLease lease = lease_device(s->device);
        (void)hipSetDevice(s->device);
        const bool pair_gone = s->pair && (s->pair->dead || !s->pair->main);  // (wedged earlier, or pool_at_exit has been here)
        if (!pair_gone && !s->wedged) {
            (void)wait_on(s, s->stream, __func__, __LINE__);
            (void)wait_on(s, s->side, __func__, __LINE__);
        }
        if (s->wedged) {
            // Work of this engine never completed on its streams: hipFree would wait for it (a device-wide synchronisation) and
            // a freed buffer could still be written.  The memory and the streams are left to the process.
            std::fprintf(stderr, "[occ] an engine whose streams never drained is closed without freeing its device memory (%s)\n", s->err.c_str());
        } else {
            destroy_graph(s);
            for (void *p : s->allocs) (void)hipFree(p);
            if (s->rec_buf) (void)hipFree(s->rec_buf);
            if (s->pin_sc) (void)hipHostFree(s->pin_sc);
            if (s->pin_rec) (void)hipHostFree(s->pin_rec);
            for (hipEvent_t ev : {s->ev0, s->ev1, s->ev_z[0], s->ev_z[1], s->ev_side[0], s->ev_side[1]})
                if (ev) (void)hipEventDestroy(ev);
        }
        drop_pair(s);  // the pair stays in the pool for the next engine of its partition (idle), unless it is dead
    }
    delete s;
    return OCC_OK;
}

// z = 1 everywhere except surveyed sites without a detection (base.py:113-119), every chain
static int init_occupancy(occ_sampler *s)
{
    const Ctx &c = s->ctx;
    std::vector<uint8_t> z0((size_t)c.C * c.n, 1);
    for (int ch = 0; ch < c.C; ++ch)
        for (int t = 0; t < c.S; ++t) z0[(size_t)ch * c.n + s->site_id[t]] = s->obs_site[t];
    HIP_TRY(copy_on(s, c.z, z0.data(), z0.size(), hipMemcpyHostToDevice));
    return OCC_OK;
}

// Everything the host derives from the caller's problem, once: checked inputs in the layouts the kernels read (SELL-64 /
// diagonal form of Q, structure-of-arrays designs, index sets of base.py:112-152, prior products).  A group of
// samplers -- one per device, or one per process -- is built from ONE layout: the root uploads it, the others receive
// the device arrays by RCCL broadcast (occ_create_group, occ_create_distributed).
struct HostLayout {
    int n = 0, S = 0, R = 0, p = 0, q = 0, ell_w = 0, rsr_dim = 0;
    double tau_rate = 0.0, tau_shape = 0.0;
    std::vector<int> sell_ptr, sell_col, dia_off, row_site, site_sidx;
    std::vector<double> sell_val, qdiag, dia_val, Xt, Wt, hyp, Kh, Qh, Eh;
    std::vector<double> prior_F;  // reference-form prior draw: n x prior_m, row-major (empty: edge form)
    int prior_m = 0;
    std::vector<uint8_t> dia_mask, yrow, obs_site;
    std::vector<int32_t> site_id, site_ptr;
};

static int build_layout(occ_sampler *s, const occ_problem *pb, HostLayout &L)
{
    if (!pb) return set_error(s, OCC_E_BADARG, "null problem");
    if (pb->n < 1 || pb->n > 0x7fffffff || pb->n_rows > 0x7fffffff || pb->n_surveyed > pb->n)
        return set_error(s, OCC_E_BADARG, "problem sizes out of range");
    if (pb->p < 1 || pb->p > OCC_MAX_COVARIATES || pb->q < 1 || pb->q > OCC_MAX_COVARIATES)
        return set_error(s, OCC_E_BADARG, "p and q must lie in [1, 32]");
    if (pb->rsr_dim > 0 && (pb->p > MAXC || pb->q > MAXC)) return set_error(s, OCC_E_BADARG, "the reduced-rank model takes at most 8 covariates of each kind");
    if (pb->rsr_dim < 0 || pb->rsr_dim > RSR_BIG_MAX || (pb->rsr_dim > 0 && (!pb->rsr_K || !pb->rsr_Q || !pb->rsr_E)))
        return set_error(s, OCC_E_BADARG, "the reduced-rank basis needs 1 to 4096 columns (rsr_K, rsr_Q, rsr_E)");
    if (!(pb->tau_rate > 0.0) || !(pb->tau_shape > 0.0)) return set_error(s, OCC_E_BADARG, "tau_rate and tau_shape must be positive");
    const int n = (int)pb->n, S = (int)pb->n_surveyed, R = (int)pb->n_rows, p = pb->p, q = pb->q;
    L.n = n; L.S = S; L.R = R; L.p = p; L.q = q; L.rsr_dim = pb->rsr_dim;
    L.tau_rate = pb->tau_rate; L.tau_shape = pb->tau_shape;
    auto &sell_ptr = L.sell_ptr; auto &sell_col = L.sell_col; auto &sell_val = L.sell_val; auto &qdiag = L.qdiag;
    auto &dia_off = L.dia_off; auto &dia_val = L.dia_val; auto &dia_mask = L.dia_mask;
    auto &Xt = L.Xt; auto &Wt = L.Wt; auto &yrow = L.yrow; auto &row_site = L.row_site; auto &site_sidx = L.site_sidx; auto &hyp = L.hyp;

    // ---- fetch and check the inputs on the host -------------------------------------------------
    std::vector<int32_t> indptr, indices;
    std::vector<double> qdata, X, W, y, a_mu, a_prec, b_mu, b_prec;
    int rc;
    if ((rc = fetch(s, indptr, pb->q_indptr, (size_t)n + 1))) return rc;
    if (indptr[0] != 0 || indptr[n] < n) return set_error(s, OCC_E_BADARG, "malformed Q indptr");
    const size_t nnz = (size_t)indptr[n];
    if ((rc = fetch(s, indices, pb->q_indices, nnz))) return rc;
    if ((rc = fetch(s, qdata, pb->q_data, nnz))) return rc;
    if ((rc = fetch(s, X, pb->X, (size_t)n * p))) return rc;
    if ((rc = fetch(s, L.site_id, pb->site_id, (size_t)S))) return rc;
    if ((rc = fetch(s, L.site_ptr, pb->site_ptr, (size_t)S + 1))) return rc;
    if ((rc = fetch(s, W, pb->W, (size_t)R * q))) return rc;
    if ((rc = fetch(s, y, pb->y, (size_t)R))) return rc;
    if ((rc = fetch(s, a_mu, pb->a_mu, (size_t)q))) return rc;
    if ((rc = fetch(s, a_prec, pb->a_prec, (size_t)q * q))) return rc;
    if ((rc = fetch(s, b_mu, pb->b_mu, (size_t)p))) return rc;
    if ((rc = fetch(s, b_prec, pb->b_prec, (size_t)p * p))) return rc;
    if (S > 0 && (L.site_ptr[0] != 0 || L.site_ptr[S] != R)) return set_error(s, OCC_E_BADARG, "site_ptr does not span the rows");

    // ---- Q: CSR -> diagonal + SELL-64 off-diagonals (coalesced per-wave slices) -------------------
    // Also checks what the edge form of the prior term needs: zero row sums, non-positive
    // off-diagonals (Q = D - W), the singular ICAR precision of gibbs/base.py:166-170.
    const int nslice = (n + 63) / 64;
    sell_ptr.assign((size_t)nslice + 1, 0);
    qdiag.assign((size_t)n, 0.0);
    double scale = 0.0;
    for (int i = 0; i < n; ++i) {
        double rowsum = 0.0, rowabs = 0.0;
        int last = -1;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int j = indices[k];
            if (j < 0 || j >= n || j <= last) return set_error(s, OCC_E_BADARG, "Q columns must be sorted, unique and in range");
            last = j;
            rowsum += qdata[k];
            rowabs += std::fabs(qdata[k]);
            if (j == i) qdiag[i] = qdata[k];
            else if (qdata[k] > 0.0 && !pb->prior_factor)
                return set_error(s, OCC_E_BADARG, "Q must have non-positive off-diagonal entries (or come with a prior factor: occ_problem::prior_factor)");
        }
        scale = std::max(scale, rowabs);
        // (with a prior factor the caller has established the singularity: F F' = Q of rank < n)
        if (!pb->prior_factor && std::fabs(rowsum) > 1e-10 * std::max(rowabs, 1e-300))
            return set_error(s, OCC_E_BADARG, "Spatial precision matrix Q must be singular.");
    }
    if (!(scale > 0.0)) return set_error(s, OCC_E_BADARG, "Spatial precision matrix Q must be singular.");
    for (int sl = 0; sl < nslice; ++sl) {
        int width = 0;
        for (int i = sl * 64; i < std::min(n, sl * 64 + 64); ++i) {
            int cnt = 0;
            for (int k = indptr[i]; k < indptr[i + 1]; ++k) cnt += (indices[k] != i);
            width = std::max(width, cnt);
        }
        sell_ptr[sl + 1] = sell_ptr[sl] + width * 64;
    }
    // uniform width (ELL) when the padding it adds is small: the slice base becomes arithmetic
    {
        int wmax = 0;
        for (int sl = 0; sl < nslice; ++sl) wmax = std::max(wmax, (sell_ptr[sl + 1] - sell_ptr[sl]) / 64);
        const long long ell_slots = (long long)wmax * 64 * nslice;
        L.ell_w = (wmax > 0 && ell_slots <= (long long)(1.25 * sell_ptr[nslice]) + 64) ? wmax : 0;
        if (L.ell_w)
            for (int sl = 0; sl <= nslice; ++sl) sell_ptr[sl] = sl * wmax * 64;
    }
    // 64 spare slots: k_iter reads slot `base + lane` of a slice even when the slice has no off-diagonals
    sell_col.assign((size_t)sell_ptr[nslice] + 64, 0);
    sell_val.assign((size_t)sell_ptr[nslice] + 64, 0.0);
    for (int sl = 0; sl < nslice; ++sl) {
        const int base = sell_ptr[sl], width = (sell_ptr[sl + 1] - base) / 64;
        for (int lane = 0; lane < 64; ++lane) {
            const int i = sl * 64 + lane;
            int kk = 0;
            if (i < n)
                for (int k = indptr[i]; k < indptr[i + 1]; ++k)
                    if (indices[k] != i) {
                        sell_col[(size_t)base + kk * 64 + lane] = indices[k];
                        sell_val[(size_t)base + kk * 64 + lane] = qdata[k];
                        ++kk;
                    }
            for (; kk < width; ++kk) sell_col[(size_t)base + kk * 64 + lane] = std::min(i, n - 1);  // padding: value 0
        }
    }

    // ---- diagonal form, when the off-diagonals lie on at most NPRE diagonals with one value each (lattices) -----
    {
        std::vector<long long> offs;
        bool ok = true;
        for (int i = 0; i < n && ok; ++i)
            for (int k = indptr[i]; k < indptr[i + 1] && ok; ++k) {
                if (indices[k] == i) continue;
                const long long d = (long long)indices[k] - i;
                size_t t = 0;
                while (t < offs.size() && offs[t] != d) ++t;
                if (t == offs.size()) {
                    if (offs.size() == (size_t)NPRE) { ok = false; break; }
                    offs.push_back(d);
                    dia_val.push_back(qdata[k]);
                } else if (dia_val[t] != qdata[k]) ok = false;
            }
        if (ok && !offs.empty()) {
            std::vector<size_t> order(offs.size());
            for (size_t t = 0; t < order.size(); ++t) order[t] = t;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return offs[a] < offs[b]; });  // CSR column order
            std::vector<double> v2;
            for (size_t t : order) { dia_off.push_back((int)offs[t]); v2.push_back(dia_val[t]); }
            dia_val = v2;
            dia_mask.assign((size_t)n, 0);
            for (int i = 0; i < n; ++i)
                for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
                    if (indices[k] == i) continue;
                    const int d = indices[k] - i;
                    for (size_t t = 0; t < dia_off.size(); ++t)
                        if (dia_off[t] == d) dia_mask[i] |= (uint8_t)(1u << t);
                }
        } else {
            dia_val.clear();
        }
    }

    // ---- design matrices as structure-of-arrays; ragged visits; index sets (base.py:112-152) -----
    Xt.assign((size_t)n * p, 0.0);
    Wt.assign((size_t)R * q, 0.0);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < p; ++a) Xt[(size_t)a * n + i] = X[(size_t)i * p + a];
    for (int r = 0; r < R; ++r)
        for (int a = 0; a < q; ++a) Wt[(size_t)a * R + r] = W[(size_t)r * q + a];
    yrow.assign((size_t)R, 0);
    row_site.assign((size_t)R, 0);
    site_sidx.assign((size_t)n, -1);
    L.obs_site.assign((size_t)S, 0);
    for (int t = 0; t < S; ++t) {
        const int site = L.site_id[t];
        if (site < 0 || site >= n || site_sidx[site] != -1) return set_error(s, OCC_E_BADARG, "site_id entries must be unique and in [0, n)");
        if (L.site_ptr[t + 1] < L.site_ptr[t]) return set_error(s, OCC_E_BADARG, "site_ptr must be non-decreasing");
        site_sidx[site] = t;
        uint8_t any = 0;
        for (int r = L.site_ptr[t]; r < L.site_ptr[t + 1]; ++r) {
            yrow[r] = (y[r] != 0.0) ? 1 : 0;
            any |= yrow[r];
        }
        L.obs_site[t] = any;
        for (int r = L.site_ptr[t]; r < L.site_ptr[t + 1]; ++r) row_site[r] = site | (any ? (int)0x80000000 : 0);
    }
    hyp.assign((size_t)q * q + q + (size_t)p * p + p, 0.0);
    {
        double *ap = hyp.data(), *apm = ap + q * q, *bp = apm + q, *bpm = bp + p * p;
        std::copy(a_prec.begin(), a_prec.end(), ap);
        std::copy(b_prec.begin(), b_prec.end(), bp);
        for (int a = 0; a < q; ++a)
            for (int b = 0; b < q; ++b) apm[a] += a_prec[(size_t)a * q + b] * a_mu[b];  // base.py:161
        for (int a = 0; a < p; ++a)
            for (int b = 0; b < p; ++b) bpm[a] += b_prec[(size_t)a * p + b] * b_mu[b];  // base.py:162
    }

    if (pb->prior_factor) {  // the reference's form of the prior draw: u = F eps
        if (pb->prior_factor_cols < 1 || pb->prior_factor_cols > n) return set_error(s, OCC_E_BADARG, "prior_factor_cols must lie in [1, n]");
        if (pb->rsr_dim > 0) return set_error(s, OCC_E_BADARG, "the reduced-rank model draws its prior term from rsr_E: no prior_factor");
        L.prior_m = (int)pb->prior_factor_cols;
        if ((rc = fetch(s, L.prior_F, pb->prior_factor, (size_t)n * L.prior_m))) return rc;
    }
    if (pb->rsr_dim > 0) {  // reduced-rank model: the basis K (n x m), K'QK and its eigenfactor (m x m), row-major
        const int m = pb->rsr_dim;
        if ((rc = fetch(s, L.Kh, pb->rsr_K, (size_t)n * m))) return rc;
        if ((rc = fetch(s, L.Qh, pb->rsr_Q, (size_t)m * m))) return rc;
        if ((rc = fetch(s, L.Eh, pb->rsr_E, (size_t)m * m))) return rc;
    }
    return OCC_OK;
}

static int create_impl(occ_sampler *s, const HostLayout &L, int32_t n_chains, const uint64_t *keys)
{
    if (!keys || n_chains < 1) return set_error(s, OCC_E_BADARG, "bad keys / n_chains");
    struct { int rsr_dim; } pbv = {L.rsr_dim}, *pb = &pbv;  // (the body below reads pb->rsr_dim)
    DeviceLease lease = lease_device(s->device);  // probes and uploads run on streams other engines of the device share
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (s->device < 0 || s->device >= ndev) return set_error(s, OCC_E_HIP, "no such HIP device");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    for (int e = 0; e < 2; ++e) {
        HIP_TRY(hipEventCreateWithFlags(&s->ev_z[e], hipEventDisableTiming | hipEventReleaseToDevice));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_side[e], hipEventDisableTiming | hipEventReleaseToDevice));
    }
    s->side_enabled = std::getenv("OCC_NO_SIDE_STREAM") == nullptr;
    if (const char *zd = std::getenv("OCC_DEBUG_ZOB_SKIP")) s->zob_debug = (std::atoi(zd) & 3) << 3;
    s->event_nodes = s->side_enabled && std::getenv("OCC_STREAM_EVENTS") == nullptr;  // diagnostic: fork/join by stream calls

    const int n = L.n, S = L.S, R = L.R, p = L.p, q = L.q, C = n_chains;
    Ctx &c = s->ctx;
    c.n = n; c.S = S; c.R = R; c.p = p; c.q = q; c.C = C;
    c.tau_rate = L.tau_rate; c.tau_shape = L.tau_shape;
    c.maxiter = 10LL * n;  // scipy default 5 * (2n)  (minres.py, called at logit.py:87)
    c.ell_w = L.ell_w;
    s->site_id = L.site_id; s->site_ptr = L.site_ptr; s->obs_site = L.obs_site;
    const auto &sell_ptr = L.sell_ptr; const auto &sell_col = L.sell_col; const auto &sell_val = L.sell_val; const auto &qdiag = L.qdiag;
    const auto &dia_off = L.dia_off; const auto &dia_val = L.dia_val; const auto &dia_mask = L.dia_mask;
    const auto &Xt = L.Xt; const auto &Wt = L.Wt; const auto &yrow = L.yrow; const auto &row_site = L.row_site;
    const auto &site_sidx = L.site_sidx; const auto &hyp = L.hyp;
    const int nslice = (n + 63) / 64;
    const uint8_t *dia_mask_dev = nullptr;
    int rc;

    // ---- launch geometry: one site (or visit row) per thread; enough blocks to spread over the CUs
    int tpb = 256;
    while (tpb > 64 && ((long long)n * C + tpb - 1) / tpb < 512) tpb >>= 1;
    // Fused iteration kernel (occ_iter.hpp): every workgroup of every chain must be resident at once (at most two
    // per CU) and a matrix row must fit the register-resident neighbour window.  Its partial sums are per
    // 64-site slice, so the other kernels use 64-thread blocks too.
    {
        int wmax = 0;
        for (int sl = 0; sl < nslice; ++sl) wmax = std::max(wmax, (sell_ptr[sl + 1] - sell_ptr[sl]) / 64);
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        const int nbg = (n + ITER_WG - 1) / ITER_WG;
        s->iter.nbg = nbg;
        // (k_iter's 240 VGPRs allow two of its workgroups per CU: 8 chains at 100x100 run 210 us per iteration that way
        // against 251 us with one launch per MINRES step)
        s->iter_window = wmax <= 8 ? 8 : 16;
        const int wg_per_cu = s->iter_window == 8 ? 2 : 1;  // 255 and ~400 VGPRs
        s->generic = p > MAXC || q > MAXC;
        const bool fused_shape = pb->rsr_dim == 0 && wmax <= 16 && !s->generic;
        const bool fused_ok = !std::getenv("OCC_NO_PERSISTENT") && fused_shape;
        s->persistent = fused_ok && (long long)nbg * C <= (long long)wg_per_cu * prop.multiProcessorCount;
        // k_tiles for what k_iter cannot hold (more sites than 64 workgroups of 512 per chain, more workgroups than two per
        // CU): T tiles of 256 sites per workgroup, four workgroups per CU (128 registers; LDS: 16 KB per tile) on HALF the
        // device -- the Polya-Gamma kernels of the side stream need the other half at these sizes (1.25 M draws per
        // iteration at 500x500) -- and at most 512 workgroups per chain (a band's records: one per lane).  The
        // LAYOUT (256-thread blocks, sums grouped by T) is decided by the shape alone, so that OCC_NO_PERSISTENT=1 and a
        // run-time fallback run the launch-per-step kernels in the same summation order: same bits.
        {
            const int ntile = (n + TILE - 1) / TILE, main_t = tiles_main_cus(prop.multiProcessorCount);
            const char *ft = std::getenv("OCC_FORCE_TILES");  // tests: 1 .. 4 = that many tiles per workgroup whatever the size
            int T = 0;
            for (int t : {1, 2, 4, 3}) {  // the fewest tiles per workgroup whose workgroups are all resident (4 before 3: it keeps its registers)
                if (T != 0) break;
                const int g = (ntile + t - 1) / t;
                if ((long long)C * g <= (long long)tiles_wg_per_cu(t) * main_t && g <= 512) T = t;
            }
            if (ft && std::atoi(ft) >= 1 && std::atoi(ft) <= 4) T = std::atoi(ft);
            // (beyond 64 workgroups of 512 sites per chain k_iter only has its any-placement form, every exchange a round trip to
            // the memory side: 250x250 x 1 chain 186 us per iteration against 123 with tiles, x 2 chains 322 / 161, 350x350
            // 299 / 152; at 150x150 x 2 chains k_iter still wins, 117 / 136)
            const bool big = n > XL_MAX_WG * ITER_WG_XL && ((long long)nbg * C > 2LL * prop.multiProcessorCount || (long long)n * C >= 50000);
            s->tiles_layout = fused_shape && wmax <= 8 && T > 0 && !std::getenv("OCC_NO_TILES") && (big || ft != nullptr);
            if (s->tiles_layout) {
                s->tiles_T = T;
                s->tiles_G = (ntile + T - 1) / T;
                s->tiles_B = (s->tiles_G + XL_SLOTS - 1) / XL_SLOTS;
                s->persistent = false;  // (k_iter's forms are out: decided below)
            }
        }
        // one XCD per chain (k_iter<8, 1, *>); candidates -- the probe below decides.  Per XCD the main stream has 20 CUs
        // (24 for larger lattices, 28 when few chains leave the side stream little to do), whole shader engines'
        // worth: an XCD deals a chain's workgroups round-robin over its four shader engines, so its CUs in the mask
        // must be a multiple of four (26 workgroups on 26 CUs per XCD dead-locked, on 28 they run).
        //   A  256-thread workgroups, one per CU            nbg <= 20
        //   B  512-thread workgroups (scalar wave + 448 sites), one per CU: ceil(n / 448) <= 28 CUs of the chain's XCD
        //   C  256-thread workgroups, two per CU            nbg <= 64 (partition of at most 24 CUs per XCD, else none)
        // Form B with fewer than eight chains: only the XCDs that host a chain need that many CUs -- the others give the
        // main stream fewer, so that the side stream keeps its share of the device (4 chains at 100x100: 24 CUs on four
        // XCDs, 16 on the other four: 160 + 96 as before).
        {
            const int ncu = prop.multiProcessorCount, nbg512 = (n + ITER_SITES_SW - 1) / ITER_SITES_SW;  // (one wave of the 512 threads owns no sites)
            const int nbg512p = (n + ITER_WG_XL - 1) / ITER_WG_XL;                                        // (eight site waves)
            const int base = (ncu * 5 / 64) * 8;  // 160 of 256
            // (more than eight chains: launches of eight, one behind the other -- 16 chains at 100x100 then run 2 x 60 us where
            // the launch-per-step path took 374)
            // (rows of 9-16 off-diagonals -- the irregular graph of BASELINE config 5 -- take the one-XCD form too, in its
            // 256-thread, one-workgroup-per-CU shape (round 3): every step's exchange through one L2 instead of the memory side)
            const bool xl_any = fused_ok && C <= 8 * XL_SLOTS && !std::getenv("OCC_NO_XCD_LOCAL");
            const bool xl_ok = xl_any && s->iter_window == 8;
            auto part = [&](int per_xcd) { return std::max(32 * ((per_xcd + 3) / 4), base); };
            s->xl_candidate = false;
            for (int x = 0; x < XL_SLOTS; ++x) s->xl_per_xcd[x] = 0;
            const int need = 4 * ((nbg512 + 3) / 4), hot = std::min(C, XL_SLOTS), per_xcd = ncu / XL_SLOTS;
            int wide_main = 0, wide_xcd[XL_SLOTS];
            if (need <= per_xcd - 4) {  // the hot XCDs leave the side stream one CU per shader engine at least
                int rest = need;
                if (hot < XL_SLOTS) {
                    rest = 4 * (int)std::lround((double)(base - hot * need) / (4.0 * (XL_SLOTS - hot)));
                    rest = std::max(8, std::min(rest, need));
                    while (hot * need + (XL_SLOTS - hot) * rest > ncu - 96 && rest > 8) rest -= 4;  // the side stream keeps 96 CUs
                    if (const char *cc = std::getenv("OCC_DEBUG_COLD_CUS")) rest = std::max(4, std::min(std::atoi(cc) / 4 * 4, need));  // developer knob
                }
                for (int x = 0; x < XL_SLOTS; ++x) { wide_xcd[x] = x < hot ? need : rest; wide_main += wide_xcd[x]; }
            }
            if (xl_any && nbg <= base / XL_SLOTS) {
                s->xl_candidate = true; s->xl_wide = 0; s->xl_nbg = nbg; s->xl_per_cu = 1; s->xl_main = base;
            } else if (xl_ok && nbg512 <= 64 && wide_main > 0 && wide_main <= ncu - 96 && hot <= 5 && !std::getenv("OCC_NO_SCALAR_WAVE")) {
                // (the scalar wave's seventh of the sites costs CUs: taken while the side stream keeps its 96 and most of them on
                // XCDs without a chain -- 100x100: 4 chains 70.0 us per iteration against 80.0 with eight site waves, 5 chains
                // 80.3 / 81.8, 6 chains 99.3 / 82.3; 8 chains on 192 + 64 CUs 121.7 / 92.5, side-stream bound)
                s->xl_candidate = true; s->xl_wide = 1; s->xl_nbg = nbg512; s->xl_per_cu = 1; s->xl_main = wide_main;
                for (int x = 0; x < XL_SLOTS; ++x) s->xl_per_xcd[x] = wide_xcd[x];
            } else if (xl_ok && nbg512p <= 64 && (part(nbg512p) <= ncu - 64 || (part(nbg512p) <= ncu - 32 && C <= 2))) {
                s->xl_candidate = true; s->xl_wide = 2; s->xl_nbg = nbg512p; s->xl_per_cu = 1; s->xl_main = part(nbg512p);
            } else if (xl_ok && nbg <= 64 && nbg <= 2 * (ncu / XL_SLOTS)) {
                s->xl_candidate = true; s->xl_wide = 0; s->xl_nbg = nbg; s->xl_per_cu = 2;
                s->xl_main = part((nbg + 1) / 2) <= ncu - 64 ? part((nbg + 1) / 2) : 0;  // 0: no CU partition
            }
        }
        s->fused_fallback = s->persistent;  // what holds without the XCD-local form
        s->persistent = s->persistent || s->xl_candidate;
        if (s->tiles_layout) {
            s->xl_candidate = false;
            s->fused_fallback = false;
            s->tiles = fused_ok;
            s->persistent = s->tiles;
            tpb = TILE;
        }
        s->tpb_plain = tpb;  // what the launch-per-step path takes when no fused form applies
        if (s->persistent && !s->tiles) tpb = 64;
    }
    // ---- streams.  The main stream carries the critical path (the eta solve); omega_a / alpha / noise of the
    // same iteration run beside it on the side stream.  With the fused iteration kernel the two streams get
    // DISJOINT sets of CUs: k_iter's workgroups are latency-bound with one wave per SIMD, and Polya-Gamma waves
    // sharing their SIMDs (long quarter-rate instructions, another kernel's code in the instruction cache) cost
    // the solve more than the side work gains from the extra CUs (100x100, 4 chains: 143 -> 127 us per
    // iteration).  A mask of N bits enables N CUs spread evenly over the 8 XCDs (tools/xcc_probe3.hip), and the
    // k_iter grid is dealt round-robin over the XCDs: a multiple of 8 keeps one workgroup per CU.
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        const int ncu = prop.multiProcessorCount;
        // at least 5/8 of the device for the main stream: k_z_ob's Polya-Gamma draws run there too
        int nmain = std::max(((s->iter.nbg * C + 7) / 8) * 8, (ncu * 5 / 64) * 8);
        // more workgroups than the partition can give one CU each: the 8-wide window runs two per CU
        if (nmain > ncu - 32 && s->iter_window == 8) nmain = std::max((((s->iter.nbg * C + 1) / 2 + 7) / 8) * 8, (ncu * 5 / 64) * 8);
        // one XCD per chain (decided for good by the probe below): a chain's nbg workgroups share the nmain / 8 CUs of
        // one XCD whatever the number of chains, two per CU
        if (s->xl_candidate) nmain = s->xl_main;  // 0: none
        if (pb->rsr_dim > 0) nmain = ((ncu * 3 / 4) / 8) * 8;  // reduced-rank model: k_rsr_gram's tiles and the theta solve
        // ... with a large basis (the m x m system in device memory: k_rsr_gram32, k_rsrb_*) the main sequence is milliseconds of
        // device-filling kernels and the side sequence 30 us: no partition, one stream (round 3 kept 64 CUs for a side stream
        // whose k_omega_a spent 4.3 ms of a 4.4 ms iteration waiting at its gate)
        if (pb->rsr_dim > RSR_MAX_DIM) nmain = 0;
        if (s->tiles) nmain = tiles_main_cus(ncu);             // k_tiles: eight tiles per CU
        if (const char *split = std::getenv("OCC_CU_SPLIT")) {  // developer knob: CUs of the main stream; 0: no masks
            nmain = std::atoi(split);
            // a partition is cut in whole shader engines per XCD (see above): multiples of 32 CUs, both streams non-empty
            if (nmain != 0 && (nmain < 32 || nmain % 32 != 0 || nmain > ncu - 32))
                return set_error(s, OCC_E_BADARG, "OCC_CU_SPLIT must be 0 (no partition) or a multiple of 32 that leaves the side stream at least 32 CUs");
        }
        if ((s->persistent || pb->rsr_dim > 0) && s->side_enabled && nmain >= 8 && nmain <= ncu - 32) {
            // bit i of a mask is CU i / 8 of XCD i % 8: the main stream takes the first per[x] CUs of XCD x
            std::vector<uint32_t> m_main((ncu + 31) / 32, 0u), m_side((ncu + 31) / 32, 0u);
            int per[XL_SLOTS];
            for (int x = 0; x < XL_SLOTS; ++x) per[x] = (s->xl_candidate && s->xl_per_xcd[0] > 0 && !std::getenv("OCC_CU_SPLIT")) ? s->xl_per_xcd[x] : nmain / XL_SLOTS;
            for (int i = 0; i < ncu; ++i) (i / XL_SLOTS < per[i % XL_SLOTS] ? m_main : m_side)[i / 32] |= 1u << (i % 32);
            s->main_hot_cus = per[0];
            // the XCDs' shares of each stream's CUs (tile_of_block_shared), when they differ; the tile tables follow
            // once the block sizes are known
            c.share_on = 0;
            s->share_cum[0][0] = s->share_cum[1][0] = 0;
            for (int x = 0; x < XL_SLOTS; ++x) {
                s->share_cum[0][x + 1] = s->share_cum[0][x] + per[x];
                s->share_cum[1][x + 1] = s->share_cum[1][x] + (ncu / XL_SLOTS - per[x]);
                if (per[x] != per[0]) c.share_on = 1;
            }
            if (std::getenv("OCC_NO_XCD_SHARES")) c.share_on = 0;
            // the pooled pair with this partition; a runtime that cannot mask CUs, or a device that has its share of masked
            // pairs already, gets the unpartitioned streams below
            if (StreamPair *pr = acquire_pair(s->device, m_main, m_side, &s->pair_note)) {
                adopt_pair(s, pr);
                s->pair_note.clear();
                s->pref.m_main = m_main;
                s->pref.m_side = m_side;
                s->main_cus = nmain;
                s->flag_sync = std::getenv("OCC_EVENT_SYNC") == nullptr;  // diagnostic: hand-overs by event nodes
                if (s->flag_sync) {
                    bool beside = true;
                    if ((rc = stream_probe(s, &beside))) return rc;
                    if (!beside) {
                        s->flag_sync = false;  // hand-overs by event nodes (ICAR) / everything on one stream (reduced-rank model)
                        s->streams_serialised = true;
                        s->probe_wait = s->probe_backoff = 1;  // (asked again at the head of the next call: refresh_paths)
                    }
                }
            } else {
                c.share_on = 0;
                if (std::getenv("OCC_VERBOSE")) std::fprintf(stderr, "[occ] no CU partition for this engine: %s\n", s->pair_note.c_str());
            }
        }
        if (s->main_cus == 0 && (rc = create_plain_streams(s))) return rc;
    }
    // ---- which form of the fused iteration kernel?  Arithmetic first (workgroups against the CUs the main stream
    // owns); the RESIDENCY PROBE further down -- k_iter itself, one barrier per chain -- has the last word.
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        const int cus = s->main_cus > 0 ? s->main_cus : prop.multiProcessorCount;
        s->any_fits = s->fused_fallback && (long long)s->iter.nbg * C <= (long long)(s->iter_window == 8 ? 2 : 1) * cus;
        s->nbg_any = s->iter.nbg;
        const bool trust = std::getenv("OCC_DEBUG_SKIP_RESIDENCY_PROBE") != nullptr;  // tests of the run-time fallback
        const int hot_cus = s->main_cus > 0 ? s->main_hot_cus : prop.multiProcessorCount / XL_SLOTS;  // CUs of a chain's XCD
        if (!trust && s->xl_candidate && s->xl_nbg > s->xl_per_cu * hot_cus) s->xl_candidate = false;
        if (s->tiles && !trust && (long long)s->tiles_G * C > (long long)tiles_wg_per_cu(s->tiles_T) * cus) s->tiles = false;
        s->persistent = s->xl_candidate || s->any_fits || s->tiles;
        if (!s->persistent) {
            tpb = s->tpb_plain;
            if (pb->rsr_dim == 0 && (rc = demote_streams(s))) return rc;
        }
    }
    s->tpb = tpb;
    c.nb_n = (n + tpb - 1) / tpb;
    s->beta_split = tpb != 64 && c.nb_n >= 128 && !std::getenv("OCC_NO_BETA_SPLIT");
    c.nb_r = std::max(1, (R + tpb - 1) / tpb);
    if (c.share_on) {  // tiles of the device-filling kernels per XCD, in proportion to the CUs of their stream
        const int per_chain[3] = {tpb == 64 ? 2 * ((n + 255) / 256) : 2 * c.nb_n, c.nb_r, (n + 255) / 256}, which[3] = {0, 1, 1};
        c.surplus_last = 0;
        if (const char *sl = std::getenv("OCC_DEBUG_SURPLUS_LAST")) c.surplus_last = std::atoi(sl);
        if (const char *cw = std::getenv("OCC_DEBUG_MAIN_SHARE")) {  // developer knob: weight of an XCD without a chain in k_z_ob's shares
            const int w = std::atoi(cw), hot = std::min(C, XL_SLOTS);
            for (int x = hot; x < XL_SLOTS; ++x) s->share_cum[0][x + 1] = s->share_cum[0][x] + w;
        }
        for (int k = 0; k < 3; ++k) {
            const long long T = (long long)per_chain[k] * C, W = s->share_cum[which[k]][8];
            c.tile_most[k] = 0;
            for (int x = 0; x <= XL_SLOTS; ++x) c.tile_first[k][x] = (int)(T * s->share_cum[which[k]][x] / W);
            for (int x = 0; x < XL_SLOTS; ++x) c.tile_most[k] = std::max(c.tile_most[k], c.tile_first[k][x + 1] - c.tile_first[k][x]);
        }
    }

    // ---- device memory ------------------------------------------------------------------------------
    if ((rc = upload(s, &c.sell_ptr, sell_ptr, "sell_ptr"))) return rc;
    if ((rc = upload(s, &c.sell_col, sell_col, "sell_col"))) return rc;
    if ((rc = upload(s, &c.sell_val, sell_val, "sell_val"))) return rc;
    if ((rc = upload(s, &c.qdiag, qdiag, "qdiag"))) return rc;
    if (!dia_off.empty() && (rc = upload(s, &dia_mask_dev, dia_mask, "dia_mask_dev"))) return rc;
    if ((rc = upload(s, &c.Xt, Xt, "Xt"))) return rc;
    if ((rc = upload(s, &c.Wt, Wt, "Wt"))) return rc;
    if ((rc = upload(s, &c.yrow, yrow, "yrow"))) return rc;
    if ((rc = upload(s, &c.row_site, row_site, "row_site"))) return rc;
    if ((rc = upload(s, &c.site_sidx, site_sidx, "site_sidx"))) return rc;
    {
        std::vector<int> sp(s->site_ptr.begin(), s->site_ptr.end());
        if ((rc = upload(s, &c.site_ptr, sp, "site_ptr"))) return rc;
    }
    if ((rc = upload(s, &c.obs_site, s->obs_site, "obs_site"))) return rc;
    if ((rc = upload(s, &c.hyp, hyp, "hyp"))) return rc;
    c.dense_F = nullptr;
    c.dense_m = 0;
    if (L.prior_m > 0) {  // reference-form prior draw
        if ((rc = upload(s, &c.dense_F, L.prior_F, "dense_F"))) return rc;
        c.dense_m = L.prior_m;
        for (int b = 0; b < 2; ++b)
            if ((rc = dev_alloc(s, &c.dense_eps[b], (size_t)C * L.prior_m))) return rc;
    }

    const size_t Cn = (size_t)C * n;
    if ((rc = dev_alloc(s, &c.eta, Cn))) return rc;
    for (int b = 0; b < 2; ++b) {
        if ((rc = dev_alloc(s, &c.omega_b[b], Cn))) return rc;
        if ((rc = dev_alloc(s, &c.enorm[b], Cn))) return rc;
        if ((rc = dev_alloc(s, &c.uprior[b], Cn))) return rc;
    }
    if ((rc = dev_alloc(s, &c.rhs, Cn))) return rc;
    if ((rc = dev_alloc(s, &c.omega_a, (size_t)C * R))) return rc;
    if ((rc = dev_alloc(s, &c.z, Cn))) return rc;
    for (int b = 0; b < 2; ++b) {
        if ((rc = dev_alloc(s, &c.Gv[b], Cn))) return rc;
        if ((rc = dev_alloc(s, &c.Wv[b], Cn))) return rc;
    }
    for (int b = 0; b < 3; ++b)
        if ((rc = dev_alloc(s, &c.Pv[b], Cn))) return rc;
    if ((rc = dev_alloc(s, &c.Xv, Cn))) return rc;
    if ((rc = dev_alloc(s, &c.part_quad, (size_t)C * c.nb_n))) return rc;
    if ((rc = dev_alloc(s, &c.part_kry, (size_t)C * 2 * 4 * c.nb_n))) return rc;
    if ((rc = dev_alloc(s, &c.part_proj, (size_t)C * 2 * c.nb_n))) return rc;
    if ((rc = dev_alloc(s, &c.part_beta, (size_t)C * nacc(p) * c.nb_n))) return rc;
    if ((rc = dev_alloc(s, &c.part_alpha, (size_t)C * nacc(q) * c.nb_r))) return rc;
    if ((rc = dev_alloc(s, &c.slots, (size_t)C * NSLOT))) return rc;
    if ((rc = dev_alloc(s, &c.sc, (size_t)C))) return rc;
    c.rec = nullptr;
    c.bar = nullptr;
    c.claim = nullptr;
    c.iter_clock = nullptr;
    c.sync = nullptr;
    if (s->main_cus > 0) {  // a CU partition: the counters exist even while events hand over (the mode can change at run time)
        if ((rc = dev_alloc(s, &s->sync_buf, (size_t)SYNC_WORDS))) return rc;
        if (std::getenv("OCC_DEBUG_BREAK_HANDOVER")) {  // tests of the run-time fallback: the side stream never announces its noise
            const unsigned one = 1u;
            HIP_TRY(copy_on(s, s->sync_buf + SYNC_DEBUG, &one, sizeof(one), hipMemcpyHostToDevice));
        }
    }
    if (s->flag_sync) {
        c.sync = s->sync_buf;
        s->iter.sync = c.sync;
    }
    if (s->tiles) {
        const size_t npad = ((size_t)n + 7) / 8 * 8;  // (a chain's exchange buffer starts on a 128-byte line)
        s->iter.tiles_npad = (int)npad;
        if ((rc = dev_alloc(s, &s->iter.tex[0], 3 * (size_t)C * npad))) return rc;  // (one allocation: k_tiles addresses the three through one descriptor)
        s->iter.tex[1] = s->iter.tex[0] + (size_t)C * npad;
        s->iter.tex[2] = s->iter.tex[0] + 2 * (size_t)C * npad;
        if ((rc = dev_alloc(s, &s->iter.trec, (size_t)C * c.nb_n * 4))) return rc;
        if ((rc = dev_alloc(s, &s->iter.tband, (size_t)C * 3 * XL_SLOTS * 4))) return rc;
        if ((rc = dev_alloc(s, &s->iter.tflag, (size_t)C * 2 * s->tiles_G))) return rc;
        s->iter.tiles_T = s->tiles_T; s->iter.tiles_G = s->tiles_G; s->iter.tiles_B = s->tiles_B;
    }
    if (s->persistent) {
        if ((rc = dev_alloc(s, &c.iter_clock, 4))) return rc;
        s->iter.clock = c.iter_clock;
        if ((rc = dev_alloc(s, &c.bar, (size_t)C * BAR_STRIDE))) return rc;
        if ((rc = dev_alloc(s, &c.claim, (size_t)C * 16))) return rc;
        s->iter.claim = c.claim;
        if ((rc = dev_alloc(s, &s->iter.part, (size_t)C * ITER_PART_DOUBLES * c.nb_n))) return rc;
        s->iter.bar = c.bar;
    }

    // initial occupancy state (base.py:113-119) and chain keys
    if ((rc = init_occupancy(s))) return rc;
    std::vector<ChainScalars> sc((size_t)C);
    std::memset(sc.data(), 0, sizeof(ChainScalars) * sc.size());
    for (int ch = 0; ch < C; ++ch) sc[ch].key = keys[ch];
    HIP_TRY(copy_on(s, c.sc, sc.data(), sizeof(ChainScalars) * sc.size(), hipMemcpyHostToDevice));
    {
        KryArgs &k = s->kry;
        k.n = c.n; k.nb_n = c.nb_n; k.ell_w = c.ell_w; k.maxiter = c.maxiter;
        k.group_T = s->tiles_layout ? s->tiles_T : 1;
        k.group_B = s->tiles_layout ? s->tiles_B : 0;
        k.dia_n = 0;
        k.dia_mask = nullptr;
        if (!dia_off.empty() && !std::getenv("OCC_NO_DIA")) {
            k.dia_n = (int)dia_off.size();
            for (size_t d = 0; d < dia_off.size(); ++d) { k.dia_off[d] = dia_off[d]; k.dia_val[d] = dia_val[d]; }
            k.dia_mask = dia_mask_dev;
        }
        k.sell_ptr = c.sell_ptr; k.sell_col = c.sell_col; k.sell_val = c.sell_val; k.qdiag = c.qdiag;
        k.omega_b[0] = c.omega_b[0]; k.omega_b[1] = c.omega_b[1];
        for (int b = 0; b < 2; ++b) { k.Gv[b] = c.Gv[b]; k.Wv[b] = c.Wv[b]; }
        for (int b = 0; b < 3; ++b) k.Pv[b] = c.Pv[b];
        k.Xv = c.Xv; k.part_kry = c.part_kry; k.part_proj = c.part_proj; k.scs = c.sc; k.slots = c.slots;
    }
    if ((rc = dev_alloc(s, &s->ctx_dev, 1, false))) return rc;
    {
        IterArgs &t = s->iter;
        t.a = s->kry;
        t.Xt = c.Xt; t.z = c.z;
        for (int b = 0; b < 2; ++b) { t.enorm[b] = c.enorm[b]; t.uprior[b] = c.uprior[b]; }
        t.rhs = c.rhs; t.eta = c.eta; t.part_quad = c.part_quad; t.part_beta = c.part_beta;
        t.tau_rate = c.tau_rate; t.tau_shape = c.tau_shape;
        t.C = c.C; t.p = c.p; t.q = c.q;
    }
    // ---- fused iteration kernel: which form is RESIDENT?  One XCD per chain first, then any placement, else one launch
    // per MINRES step (the partial sums stay per 64-site slice: the three paths return the same bits).
    if (s->persistent) {
        const bool trust = std::getenv("OCC_DEBUG_SKIP_RESIDENCY_PROBE") != nullptr;  // tests of the run-time fallback
        bool ok = false;
        if (s->tiles) {
            ok = trust;
            if (trust) { if ((rc = tiles_reset(s))) return rc; }
            else if ((rc = residency_probe(s, &ok))) return rc;
            if (std::getenv("OCC_VERBOSE"))
                std::fprintf(stderr, "[occ] k_tiles: %d tiles per workgroup, %d workgroups per chain in bands of %d per XCD, main stream %d CUs: %s\n",
                             s->tiles_T, s->tiles_G, s->tiles_B, s->main_cus, ok ? "resident" : "NOT resident");
            if (!ok) s->tiles = false;
        }
        if (s->xl_candidate) {
            s->xcd_local = true;
            s->iter.nbg = s->xl_nbg;
            ok = trust;
            if (!trust && (rc = residency_probe(s, &ok))) return rc;
            if (std::getenv("OCC_VERBOSE"))
                std::fprintf(stderr, "[occ] one XCD per chain: %d workgroups of %d threads per chain, main stream %d CUs (%d on a chain's XCD): %s\n",
                             s->xl_nbg, s->xl_wide ? ITER_WG_XL : ITER_WG, s->main_cus, s->main_hot_cus, ok ? "resident" : "NOT resident");
            if (!ok) s->xcd_local = false;
        }
        if (!ok && s->any_fits) {
            s->xl_wide = 0;
            s->iter.nbg = s->nbg_any;
            ok = trust;
            if (!trust && (rc = residency_probe(s, &ok))) return rc;
        }
        if (!ok) {
            s->persistent = false;
            s->iter.nbg = s->nbg_any;
            if ((rc = demote_streams(s))) return rc;
            if (!s->flag_sync) { c.sync = nullptr; s->iter.sync = nullptr; }
            if (std::getenv("OCC_VERBOSE")) std::fprintf(stderr, "[occ] the fused iteration kernel is not resident on this device: one launch per MINRES step\n");
        }
    }
    if (pb->rsr_dim > 0) {  // reduced-rank model
        const int m = pb->rsr_dim;
        const std::vector<double> &Kh = L.Kh, &Qh = L.Qh, &Eh = L.Eh;
        std::vector<double> Kth((size_t)m * n);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < m; ++a) Kth[(size_t)a * n + i] = Kh[(size_t)i * m + a];
        RsrArgs &r = s->rsr;
        r.n = n; r.m = m; r.p = p; r.C = C;
        r.ldk = m > RSR_MAX_DIM ? 32 * ((m + 31) / 32) : 16 * ((m + 15) / 16);  // (large bases: k_rsr_gram32 reads 32 columns per block)
        std::vector<double> Kp((size_t)n * r.ldk, 0.0);  // rows padded to whole 128-byte lines
        for (int i = 0; i < n; ++i) std::copy(Kh.begin() + (size_t)i * m, Kh.begin() + (size_t)(i + 1) * m, Kp.begin() + (size_t)i * r.ldk);
        if ((rc = upload(s, &r.K, Kp, "rsr_K"))) return rc;
        if ((rc = upload(s, &r.Kt, Kth, "rsr_Kt"))) return rc;
        if ((rc = upload(s, &r.Qr, Qh, "rsr_Qr"))) return rc;
        std::vector<double> Eth((size_t)m * m);
        for (int a = 0; a < m; ++a)
            for (int j = 0; j < m; ++j) Eth[(size_t)j * m + a] = Eh[(size_t)a * m + j];
        if ((rc = upload(s, &r.Et, Eth, "rsr_Et"))) return rc;
        r.Xt = c.Xt; r.z = c.z;
        for (int b = 0; b < 2; ++b) { r.omega_b[b] = c.omega_b[b]; r.enorm[b] = c.enorm[b]; }
        if ((rc = dev_alloc(s, &r.theta, (size_t)C * m))) return rc;
        if ((rc = dev_alloc(s, &r.gram, (size_t)C * m * m))) return rc;
        r.nchunk = 1;
        if ((rc = dev_alloc(s, &r.rhs, (size_t)C * r.nchunk * m))) return rc;
        r.eta = c.eta;
        r.tau_rate = c.tau_rate; r.tau_shape = c.tau_shape;
        r.scs = c.sc;
        r.sync = c.sync;
        s->rsr_K_host = Kh;
        r.E = nullptr; r.big_eps = nullptr; r.big_scal = nullptr; r.big_rhs = nullptr; r.big_dfac = nullptr; r.big_quad = nullptr; r.big_u = nullptr;
        if (m <= RSR_MAX_DIM) {
            HIP_TRY(hipFuncSetAttribute((const void *)pick_rsr_solve(m), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(double) * rsr_solve_lds_doubles(m))));
        } else {  // the global-memory solve (k_rsrb_*)
            HIP_TRY(hipFuncSetAttribute((const void *)k_rsr_gram32<1, GRAM32_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsr_gram32_lds(GRAM32_WAVES)));
            HIP_TRY(hipFuncSetAttribute((const void *)k_rsr_gram32<2, GRAM32_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsr_gram32_lds(GRAM32_WAVES)));
            if ((rc = upload(s, &r.E, Eh, "rsr_E"))) return rc;
            if ((rc = dev_alloc(s, &r.big_eps, (size_t)C * m))) return rc;
            if ((rc = dev_alloc(s, &r.big_scal, (size_t)C * 2))) return rc;
            if ((rc = dev_alloc(s, &r.big_quad, (size_t)C * ((m + RSRB_QROWS - 1) / RSRB_QROWS)))) return rc;
            if ((rc = dev_alloc(s, &r.big_u, (size_t)C * n))) return rc;
            if ((rc = dev_alloc(s, &r.big_rhs, (size_t)C * m))) return rc;
            if ((rc = dev_alloc(s, &r.big_dfac, (size_t)C * ((m + RSR_PANEL - 1) / RSR_PANEL) * RSR_PANEL * RSR_PANEL))) return rc;
        }
    }
    HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    // (the engine's own two streams, not hipDeviceSynchronize: other engines' pooled streams are not this call's business)
    WAIT_TRY(s->stream);
    WAIT_TRY(s->side);
    // what the engine comes back to after a run-time fallback (try_repromote)
    s->pref.valid = true;
    s->pref.persistent = s->persistent;
    s->pref.xcd_local = s->xcd_local;
    s->pref.tiles = s->tiles;
    s->pref.flag_sync = s->main_cus > 0 && s->sync_buf != nullptr && std::getenv("OCC_EVENT_SYNC") == nullptr;
    s->pref.share_on = c.share_on;
    s->pref.main_cus = s->main_cus;
    if (s->main_cus == 0) { s->pref.m_main.clear(); s->pref.m_side.clear(); }
    return OCC_OK;
}

int occ_create(const occ_problem *problem, int32_t n_chains, const uint64_t *keys, int32_t device, occ_sampler **out)
{
    if (!out) return OCC_E_BADARG;
    *out = nullptr;
    occ_sampler *s = new occ_sampler();
    s->device = device;
    HostLayout L;
    int rc = hipSetDevice(device) == hipSuccess ? OCC_OK : set_error(s, OCC_E_HIP, "no such HIP device");
    if (rc == OCC_OK) rc = build_layout(s, problem, L);
    if (rc == OCC_OK) rc = create_impl(s, L, n_chains, keys);
    if (rc != OCC_OK) {
        g_create_error = s->err;
        occ_destroy(s);
        return rc;
    }
    *out = s;
    return OCC_OK;
}

// ---- groups of samplers: one per device (in-process) or one per process, the fixed arrays broadcast over RCCL ---------
namespace {

#define NCCL_TRY(owner, expr)                                                                                 \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            (owner)->err = std::string(#expr) + ": " + rccl().GetErrorString(r_);                             \
            return OCC_E_HIP;                                                                                 \
        }                                                                                                     \
    } while (0)

std::string g_comm_error;  // (not thread_local: the Python side may run occ_comm_create on a helper thread and ask from another)

int comm_stage(occ_comm *cm, size_t bytes)
{
    if (bytes <= cm->stage_bytes) return OCC_OK;
    if (cm->stage) (void)hipFree(cm->stage);
    cm->stage = nullptr;
    cm->stage_bytes = 0;
    if (hipMalloc(&cm->stage, bytes) != hipSuccess) { cm->err = "hipMalloc of the staging buffer failed"; return OCC_E_HIP; }
    cm->stage_bytes = bytes;
    return OCC_OK;
}

int comm_bcast_dev(occ_comm *cm, void *dev_ptr, size_t bytes, int root)
{
    if (bytes == 0) return OCC_OK;
    NCCL_TRY(cm, rccl().Broadcast(dev_ptr, dev_ptr, bytes, ncclChar, root, cm->comm, cm->stream));
    return OCC_OK;
}

int comm_bcast_host(occ_comm *cm, void *buf, size_t bytes, int root)
{
    if (bytes == 0) return OCC_OK;
    int rc = comm_stage(cm, bytes);
    if (rc) return rc;
    if (cm->rank == root && hipMemcpyAsync(cm->stage, buf, bytes, hipMemcpyHostToDevice, cm->stream) != hipSuccess) { cm->err = "H2D copy failed"; return OCC_E_HIP; }
    if ((rc = comm_bcast_dev(cm, cm->stage, bytes, root))) return rc;
    if (hipMemcpyAsync(buf, cm->stage, bytes, hipMemcpyDeviceToHost, cm->stream) != hipSuccess || hipStreamSynchronize(cm->stream) != hipSuccess) {
        cm->err = "D2H copy after the broadcast failed";
        return OCC_E_HIP;
    }
    return OCC_OK;
}

int comm_allreduce(occ_comm *cm, double *inout, int n, ncclRedOp_t op)
{
    int rc = comm_stage(cm, sizeof(double) * (size_t)n);
    if (rc) return rc;
    if (hipMemcpyAsync(cm->stage, inout, sizeof(double) * n, hipMemcpyHostToDevice, cm->stream) != hipSuccess) { cm->err = "H2D copy failed"; return OCC_E_HIP; }
    NCCL_TRY(cm, rccl().AllReduce(cm->stage, cm->stage, (size_t)n, ncclDouble, op, cm->comm, cm->stream));
    if (hipMemcpyAsync(inout, cm->stage, sizeof(double) * n, hipMemcpyDeviceToHost, cm->stream) != hipSuccess || hipStreamSynchronize(cm->stream) != hipSuccess) {
        cm->err = "D2H copy after the reduction failed";
        return OCC_E_HIP;
    }
    return OCC_OK;
}

// What a peer needs to size its arrays before the broadcast
struct LayoutHeader {
    int32_t ok, n, S, R, p, q, ell_w, ndia, nsell_ptr, pad_;
    double tau_rate, tau_shape;
    int32_t dia_off[8];
    double dia_val[8];
};

void size_peer_layout(HostLayout &L, const LayoutHeader &h)
{
    L.n = h.n; L.S = h.S; L.R = h.R; L.p = h.p; L.q = h.q; L.ell_w = h.ell_w; L.rsr_dim = 0;
    L.tau_rate = h.tau_rate; L.tau_shape = h.tau_shape;
    L.dia_off.assign(h.dia_off, h.dia_off + h.ndia);
    L.dia_val.assign(h.dia_val, h.dia_val + h.ndia);
    const size_t slots = (size_t)L.sell_ptr.back() + 64;
    L.sell_col.assign(slots, 0);
    L.sell_val.assign(slots, 0.0);
    L.qdiag.assign((size_t)h.n, 0.0);
    if (h.ndia > 0) L.dia_mask.assign((size_t)h.n, 0);
    L.Xt.assign((size_t)h.n * h.p, 0.0);
    L.Wt.assign((size_t)h.R * h.q, 0.0);
    L.yrow.assign((size_t)h.R, 0);
    L.row_site.assign((size_t)h.R, 0);
    L.site_sidx.assign((size_t)h.n, -1);
    L.hyp.assign((size_t)h.q * h.q + h.q + (size_t)h.p * h.p + h.p, 0.0);
    L.site_id.assign((size_t)h.S, 0);
    for (int t = 0; t < h.S; ++t) L.site_id[t] = t;  // placeholders (unique, in range) until the real arrays arrive
    L.site_ptr.assign((size_t)h.S + 1, 0);
    L.obs_site.assign((size_t)h.S, 0);
}

// Checksums of the sampler's fixed arrays as they sit on ITS device (k_checksum), in fixed_list order: what a group
// compares after the broadcast -- a first multi-GPU run must be able to tell a wrong broadcast from a right one.
int fixed_checksums(occ_sampler *s, std::vector<unsigned long long> &sums)
{
    const size_t na = s->fixed_list.size();
    sums.assign(na, 0ull);
    if (na == 0) return OCC_OK;
    unsigned long long *d = nullptr;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMalloc((void **)&d, sizeof(unsigned long long) * na));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long) * na, s->stream);
    for (size_t i = 0; i < na && e == hipSuccess; ++i) {
        const size_t bytes = s->fixed_list[i].second;
        const unsigned blocks = (unsigned)std::min<size_t>(1024, (bytes / 8 + 255) / 256 + 1);
        hipLaunchKernelGGL(k_checksum, dim3(blocks), dim3(256), 0, s->stream, (const unsigned char *)s->fixed_list[i].first, (unsigned long long)bytes, d + i);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(sums.data(), d, sizeof(unsigned long long) * na, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = drain_for_copy(s);
    (void)hipFree(d);
    if (e != hipSuccess) {
        s->err = std::string("checksum of the fixed arrays failed: ") + hipGetErrorString(e);
        return OCC_E_HIP;
    }
    if (const char *dbg = std::getenv("OCC_DEBUG_CORRUPT_BROADCAST")) {  // tests: as if array <n> had arrived damaged on a peer
        const size_t k = (size_t)std::atoi(dbg);
        if (s->defer_fixed && k < na) sums[k] ^= 1ull;
    }
    return OCC_OK;
}

// A peer's host mirrors (site numbers, row offsets, detection flags) from the device arrays it has just received
int refresh_host_mirrors(occ_sampler *s)
{
    const Ctx &c = s->ctx;
    std::vector<int> sidx((size_t)c.n), sp((size_t)c.S + 1);
    s->obs_site.assign((size_t)c.S, 0);
    HIP_TRY(copy_on(s, sidx.data(), c.site_sidx, sizeof(int) * sidx.size(), hipMemcpyDeviceToHost));
    HIP_TRY(copy_on(s, sp.data(), c.site_ptr, sizeof(int) * sp.size(), hipMemcpyDeviceToHost));
    if (c.S > 0) HIP_TRY(copy_on(s, s->obs_site.data(), c.obs_site, (size_t)c.S, hipMemcpyDeviceToHost));
    s->site_id.assign((size_t)c.S, 0);
    for (int i = 0; i < c.n; ++i)
        if (sidx[i] >= 0 && sidx[i] < c.S) s->site_id[sidx[i]] = i;
    s->site_ptr.assign(sp.begin(), sp.end());
    return init_occupancy(s);
}

}  // namespace

extern "C" {

const char *occ_comm_last_error(const occ_comm *cm) { return cm ? cm->err.c_str() : g_comm_error.c_str(); }

int occ_comm_unique_id(uint8_t id[128])
{
    if (!id) return OCC_E_BADARG;
    if (!rccl().ok()) { g_comm_error = rccl().err; return OCC_E_HIP; }
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r); return OCC_E_HIP; }
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &u, 128);
    return OCC_OK;
}

int occ_comm_create(int32_t world, int32_t rank, const uint8_t id[128], int32_t device, occ_comm **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return OCC_E_BADARG;
    *out = nullptr;
    if (!rccl().ok()) { g_comm_error = rccl().err; return OCC_E_HIP; }
    occ_comm *cm = new occ_comm();
    cm->world = world; cm->rank = rank; cm->device = device;
    auto fail = [&](const std::string &msg) { g_comm_error = msg; delete cm; return OCC_E_HIP; };
    if (hipSetDevice(device) != hipSuccess) return fail("no such HIP device");
    if (hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    const ncclResult_t r = rccl().CommInitRank(&cm->comm, world, u, rank);
    if (r != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    *out = cm;
    return OCC_OK;
}

int occ_comm_destroy(occ_comm *cm)
{
    if (!cm) return OCC_OK;
    (void)hipSetDevice(cm->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->comm) (void)rccl().CommDestroy(cm->comm);
    if (cm->stage) (void)hipFree(cm->stage);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
    return OCC_OK;
}

int occ_comm_barrier(occ_comm *cm)
{
    if (!cm) return OCC_E_BADARG;
    (void)hipSetDevice(cm->device);
    double one = 1.0;
    return comm_allreduce(cm, &one, 1, ncclSum);
}

int occ_comm_allreduce_max(occ_comm *cm, double *inout, int32_t n)
{
    if (!cm || !inout || n < 1) return OCC_E_BADARG;
    (void)hipSetDevice(cm->device);
    return comm_allreduce(cm, inout, n, ncclMax);
}

int occ_comm_broadcast_host(occ_comm *cm, void *buf, int64_t bytes, int32_t root)
{
    if (!cm || (!buf && bytes > 0) || bytes < 0 || root < 0 || root >= cm->world) return OCC_E_BADARG;
    (void)hipSetDevice(cm->device);
    return comm_bcast_host(cm, buf, (size_t)bytes, root);
}

// Everything enqueued on the handle's device has completed (benchmarks bracket their timed region with it).
int occ_synchronize(occ_sampler *s)
{
    if (!s) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    WAIT_TRY(s->stream);  // (bounded and self-naming; what is left for the device-wide call is other engines' work)
    WAIT_TRY(s->side);
    HIP_TRY(hipDeviceSynchronize());
    return OCC_OK;
}

const char *occ_group_transport(const occ_sampler *s) { return s ? s->group_transport.c_str() : ""; }

// One sampler per PROCESS (one process per GPU): the root lays the problem out and uploads it; every other rank sizes its
// arrays from a small header and receives the fixed arrays device-to-device over RCCL (xGMI) -- no host copy of the
// design matrices exists on the peers at any time.  `problem` is read on the root only.
int occ_create_distributed(const occ_problem *problem, occ_comm *cm, int32_t root, int32_t n_chains, const uint64_t *keys, occ_sampler **out)
{
    if (!out || !cm || root < 0 || root >= cm->world) return OCC_E_BADARG;
    *out = nullptr;
    occ_sampler *s = new occ_sampler();
    s->device = cm->device;
    auto fail = [&](int rc) { g_create_error = s->err.empty() ? cm->err : s->err; occ_destroy(s); return rc; };
    if (hipSetDevice(cm->device) != hipSuccess) return fail(set_error(s, OCC_E_HIP, "no such HIP device"));
    HostLayout L;
    LayoutHeader h{};
    int rc = OCC_OK;
    if (cm->rank == root) {
        rc = build_layout(s, problem, L);
        if (rc == OCC_OK && (L.rsr_dim > 0 || L.prior_m > 0))
            rc = set_error(s, OCC_E_BADARG, "occ_create_distributed covers the ICAR model with the edge-form prior draw");
        h.ok = rc == OCC_OK;
        if (h.ok) {
            h.n = L.n; h.S = L.S; h.R = L.R; h.p = L.p; h.q = L.q; h.ell_w = L.ell_w;
            h.ndia = (int32_t)L.dia_off.size(); h.nsell_ptr = (int32_t)L.sell_ptr.size();
            h.tau_rate = L.tau_rate; h.tau_shape = L.tau_shape;
            for (int d = 0; d < h.ndia; ++d) { h.dia_off[d] = L.dia_off[d]; h.dia_val[d] = L.dia_val[d]; }
        }
    }
    int crc = comm_bcast_host(cm, &h, sizeof(h), root);  // (a root that failed says so: nobody waits for arrays that never come)
    if (crc) return fail(crc);
    if (!h.ok) return fail(rc ? rc : set_error(s, OCC_E_BADARG, "the root rank rejected the problem"));
    if (cm->rank != root) L.sell_ptr.assign((size_t)h.nsell_ptr, 0);
    if ((crc = comm_bcast_host(cm, L.sell_ptr.data(), sizeof(int) * L.sell_ptr.size(), root))) return fail(crc);
    if (cm->rank != root) {
        size_peer_layout(L, h);
        s->defer_fixed = true;
    }
    rc = create_impl(s, L, n_chains, keys);
    // every rank holds the arrays the root is about to send, in its order and sizes (a mismatch would hang or misroute the
    // broadcasts): the root's (count, sizes[]) go round first
    {
        std::vector<long long> want(65, -1);
        if (cm->rank == root && rc == OCC_OK) {
            want[0] = (long long)s->fixed_list.size();
            for (size_t i = 0; i < s->fixed_list.size() && i < 64; ++i) want[1 + i] = (long long)s->fixed_list[i].second;
        }
        if ((crc = comm_bcast_host(cm, want.data(), sizeof(long long) * want.size(), root))) return fail(crc);
        if (rc == OCC_OK) {
            bool same = want[0] == (long long)s->fixed_list.size() && s->fixed_list.size() <= 64;
            for (size_t i = 0; same && i < s->fixed_list.size(); ++i) same = want[1 + i] == (long long)s->fixed_list[i].second;
            if (!same) rc = set_error(s, OCC_E_HIP, "internal: this rank's fixed arrays differ in number or size from the root's");
        }
    }
    double bad = rc != OCC_OK ? 1.0 : 0.0;  // all ranks or none go on to the broadcasts
    if ((crc = comm_allreduce(cm, &bad, 1, ncclMax))) return fail(crc);
    if (bad != 0.0) return fail(rc ? rc : set_error(s, OCC_E_HIP, "another rank failed to create its sampler"));
    if (hipDeviceSynchronize() != hipSuccess) bad = 1.0;  // (reported after the collectives below: nobody is left inside a broadcast)
    for (auto &fa : s->fixed_list)
        if ((crc = comm_bcast_dev(cm, fa.first, fa.second, root))) return fail(crc);
    if (hipStreamSynchronize(cm->stream) != hipSuccess) bad = 1.0;
    // ... and what arrived is what was sent: every rank checksums its copies on the device, the root's sums go round,
    // the first array that differs anywhere is named on every rank
    DeviceLease lease = lease_device(s->device);  // the checksum kernels and the mirrors' copies run on the (pooled) main stream
    {
        std::vector<unsigned long long> mine, roots;
        if (bad == 0.0 && fixed_checksums(s, mine) != OCC_OK) bad = 1.0;
        roots = mine;
        roots.resize(s->fixed_list.size(), 0ull);
        if ((crc = comm_bcast_host(cm, roots.data(), sizeof(unsigned long long) * roots.size(), root))) return fail(crc);
        double first_diff = 0.0;  // 1 + index of the first differing array over all ranks (max: any)
        for (size_t i = 0; bad == 0.0 && i < roots.size(); ++i)
            if (mine[i] != roots[i]) { first_diff = (double)(i + 1); break; }
        double flags[2] = {bad, first_diff};
        if ((crc = comm_allreduce(cm, flags, 2, ncclMax))) return fail(crc);
        if (flags[0] != 0.0) return fail(set_error(s, OCC_E_HIP, "a rank failed while the problem was broadcast"));
        if (flags[1] != 0.0) {
            const size_t k = (size_t)flags[1] - 1;
            const std::string nm = k < s->fixed_names.size() ? s->fixed_names[k] : "?";
            return fail(set_error(s, OCC_E_HIP, ("the broadcast of the problem did not arrive intact: array `" + nm + "` differs from the root's on at least one rank").c_str()));
        }
    }
    if (cm->rank != root && (rc = refresh_host_mirrors(s))) return fail(rc);
    s->group_transport = "rccl broadcast (ncclCommInitRank), " + std::to_string(cm->world) + " ranks";
    *out = s;
    return OCC_OK;
}

// One sampler per DEVICE of this process (driven by one host thread each): the problem is laid out once, uploaded to
// devices[0] and broadcast from there to the other devices -- RCCL (ncclCommInitAll + grouped ncclBroadcast), or, when
// librccl cannot be used (or OCC_GROUP_TRANSPORT=peer), hipMemcpyPeer, device to device either way.
// keys: the chains' keys, device after device.  On failure no handle is returned.
int occ_create_group(const occ_problem *problem, int32_t n_devices, const int32_t *devices, const int32_t *chains_per_device,
                     const uint64_t *keys, occ_sampler **out)
{
    if (!out || !devices || !chains_per_device || !keys || n_devices < 1) return OCC_E_BADARG;
    for (int g = 0; g < n_devices; ++g) out[g] = nullptr;
    std::vector<occ_sampler *> ss((size_t)n_devices, nullptr);
    auto fail = [&](int rc, const std::string &msg) {
        g_create_error = msg;
        for (auto *p : ss) occ_destroy(p);
        return rc;
    };
    HostLayout L;
    {
        occ_sampler tmp;
        tmp.device = devices[0];
        if (hipSetDevice(devices[0]) != hipSuccess) return fail(OCC_E_HIP, "no such HIP device");
        const int rc = build_layout(&tmp, problem, L);
        if (rc) return fail(rc, tmp.err);
    }
    size_t koff = 0;
    for (int g = 0; g < n_devices; ++g) {
        ss[g] = new occ_sampler();
        ss[g]->device = devices[g];
        ss[g]->defer_fixed = g > 0;
        if (hipSetDevice(devices[g]) != hipSuccess) return fail(OCC_E_HIP, "no such HIP device");
        const int rc = create_impl(ss[g], L, chains_per_device[g], keys + koff);
        if (rc) return fail(rc, ss[g]->err);
        if (hipDeviceSynchronize() != hipSuccess) return fail(OCC_E_HIP, "device synchronisation failed");
        koff += (size_t)chains_per_device[g];
    }
    // the hand-over below runs on the samplers' (pooled) streams: nobody else's calls in between
    std::vector<DeviceLease> leases;
    {
        std::vector<int> ds(devices, devices + n_devices);
        std::sort(ds.begin(), ds.end());
        ds.erase(std::unique(ds.begin(), ds.end()), ds.end());
        for (int d : ds) leases.push_back(lease_device(d));
    }
    const size_t narr = ss[0]->fixed_list.size();
    for (int g = 1; g < n_devices; ++g) {
        if (ss[g]->fixed_list.size() != narr) return fail(OCC_E_HIP, "internal: the samplers of a group disagree on their arrays");
        for (size_t i = 0; i < narr; ++i)
            if (ss[g]->fixed_list[i].second != ss[0]->fixed_list[i].second) return fail(OCC_E_HIP, "internal: the samplers of a group disagree on their arrays");
    }
    std::string transport = "single device";
    const char *force = std::getenv("OCC_GROUP_TRANSPORT");
    // (OCC_GROUP_TRANSPORT=rccl runs the RCCL broadcast for a group of ONE device too: an in-place broadcast among one
    // rank, which is how a one-GPU box exercises ncclCommInitAll and the grouped ncclBroadcast calls)
    if (n_devices > 1 || (force && std::string(force) == "rccl")) {
        bool done = false;
        std::string why;
        if (!(force && std::string(force) == "peer") && rccl().ok()) {
            std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
            ncclResult_t r = rccl().CommInitAll(comms.data(), n_devices, devices);
            if (r == ncclSuccess) {
                for (size_t i = 0; i < narr && r == ncclSuccess; ++i) {
                    r = rccl().GroupStart();
                    for (int g = 0; g < n_devices && r == ncclSuccess; ++g) {
                        void *ptr = ss[g]->fixed_list[i].first;
                        r = rccl().Broadcast(ptr, ptr, ss[0]->fixed_list[i].second, ncclChar, 0, comms[g], ss[g]->stream);
                    }
                    const ncclResult_t re = rccl().GroupEnd();
                    if (r == ncclSuccess) r = re;
                }
                for (int g = 0; g < n_devices; ++g) {
                    (void)hipSetDevice(devices[g]);
                    if (hipStreamSynchronize(ss[g]->stream) != hipSuccess && r == ncclSuccess) r = ncclUnhandledCudaError;
                }
                for (auto cmm : comms)
                    if (cmm) (void)rccl().CommDestroy(cmm);
                if (r != ncclSuccess) return fail(OCC_E_HIP, std::string("RCCL broadcast of the problem failed: ") + rccl().GetErrorString(r));
                done = true;
                transport = "rccl broadcast (ncclCommInitAll), " + std::to_string(n_devices) + " devices";
            } else {
                why = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r);
                (void)hipGetLastError();
            }
        } else {
            why = force ? "OCC_GROUP_TRANSPORT=peer" : rccl().err;
        }
        if (!done) {  // device-to-device copies from the root
            for (int g = 1; g < n_devices; ++g)
                for (size_t i = 0; i < narr; ++i)
                    if (hipMemcpyPeer(ss[g]->fixed_list[i].first, devices[g], ss[0]->fixed_list[i].first, devices[0], ss[0]->fixed_list[i].second) != hipSuccess)
                        return fail(OCC_E_HIP, "hipMemcpyPeer of the problem failed (" + why + ")");
            transport = "hipMemcpyPeer (" + why + ")";
        }
    }
    // what arrived is what was sent: every sampler checksums its copies on its own device; a difference names the array
    if (n_devices > 1 || (force && std::string(force) == "rccl")) {
        std::vector<unsigned long long> root_sums, sums;
        if (fixed_checksums(ss[0], root_sums) != OCC_OK) return fail(OCC_E_HIP, ss[0]->err);
        for (int g = 1; g < n_devices; ++g) {
            if (fixed_checksums(ss[g], sums) != OCC_OK) return fail(OCC_E_HIP, ss[g]->err);
            for (size_t i = 0; i < narr; ++i)
                if (sums[i] != root_sums[i])
                    return fail(OCC_E_HIP, std::string("the broadcast of the problem did not arrive intact: array `") + ss[0]->fixed_names[i] + "` on device " +
                                               std::to_string(devices[g]) + " (sampler " + std::to_string(g) + ") differs from the root's (" + transport + ")");
        }
    }
    for (int g = 0; g < n_devices; ++g) {
        ss[g]->group_transport = transport;
        out[g] = ss[g];
    }
    return OCC_OK;
}

}  // extern "C"

// theta of one chain (caller pointer), and eta = K theta computed on the host (set-up path only)
static int set_theta(occ_sampler *s, int chain, const double *theta_in)
{
    const int n = s->ctx.n, m = s->rsr.m;
    std::vector<double> th;
    int rc = fetch(s, th, theta_in, (size_t)m);
    if (rc) return rc;
    std::vector<double> eta((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int a = 0; a < m; ++a) t = std::fma(s->rsr_K_host[(size_t)i * m + a], th[a], t);
        eta[i] = t;
    }
    HIP_TRY(copy_on(s, s->rsr.theta + (size_t)chain * m, th.data(), sizeof(double) * m, hipMemcpyHostToDevice));
    HIP_TRY(copy_on(s, s->ctx.eta + (size_t)chain * n, eta.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    return OCC_OK;
}

int occ_set_start(occ_sampler *s, int32_t chain, const double *alpha, const double *beta, double tau, const double *eta)
{
    if (!s) return OCC_E_BADARG;
    const Ctx &c = s->ctx;
    if (chain < 0 || chain >= c.C || !alpha || !beta || !eta) return set_error(s, OCC_E_BADARG, "bad chain / null start pointer");
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<ChainScalars> h;
    int rc = read_scalars(s, h);
    if (rc) return rc;
    ChainScalars &sc = h[chain];
    std::vector<double> a, b;
    if ((rc = fetch(s, a, alpha, (size_t)c.q))) return rc;
    if ((rc = fetch(s, b, beta, (size_t)c.p))) return rc;
    std::memset(sc.alpha, 0, sizeof(sc.alpha));
    std::memset(sc.beta, 0, sizeof(sc.beta));
    std::copy(a.begin(), a.end(), sc.alpha);
    std::copy(b.begin(), b.end(), sc.beta);
    sc.tau = tau;
    const Ctl fresh = {0u, 0u};
    sc.ctl[0] = sc.ctl[1] = sc.mid[0] = sc.mid[1] = fresh;
    sc.it_stop = 0; sc.it_base = 0; sc.burnin = 0; sc.keep = 0; sc.err = 0;
    if ((rc = write_scalars(s, h))) return rc;
    if (s->rsr.m > 0) {  // `eta` holds theta; the spatial effects follow (logit.py:457-460)
        if ((rc = set_theta(s, chain, eta))) return rc;
    } else {
        HIP_TRY(copy_on(s, c.eta + (size_t)chain * c.n, eta, sizeof(double) * c.n, hipMemcpyDefault));
    }
    HIP_TRY(fill_on(s, c.Xv + (size_t)chain * c.n, 0, sizeof(double2) * c.n));  // x0 = None (logit.py:71)
    s->need_prologue = true;
    return OCC_OK;
}

int occ_set_keys(occ_sampler *s, const uint64_t *keys)
{
    if (!s || !keys) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<ChainScalars> h;
    int rc = read_scalars(s, h);
    if (rc) return rc;
    for (size_t c = 0; c < h.size(); ++c) h[c].key = keys[c];
    s->need_prologue = true;
    return write_scalars(s, h);
}

static int step_impl(occ_sampler *s, bool snapshot)
{
    int rc = open_window(s, 1, 0, 0, snapshot, false);
    if (rc) return rc;
    if ((rc = eager_sequence(s))) return rc;
    std::vector<ChainScalars> h;
    if ((rc = read_scalars(s, h))) return rc;
    s->iterations = h[0].ctl[s->parity].it;
    s->krylov_last = h[0].minres_itn_last;
    if ((rc = check_device_errors(s, h))) return rc;
    s->mirror = h;
    s->mirror_ok = true;
    s->clean_exit = true;
    return OCC_OK;
}

// Behind the batch that is expected to finish the call: the end-of-run event and the copy of the recorded rows, so that the
// one synchronisation of the scalars' read covers them (a batch that turns out NOT to be the last -- a carried solve on the
// launch-per-step path -- simply gets them again behind the next one).
static int finish_marks(occ_sampler *s, bool last, size_t n_rec)
{
    if (!last) return OCC_OK;
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    if (n_rec > s->pin_rec_cap) {  // (grown between calls only: nothing is in flight to the old one)
        if (s->pin_rec) HIP_TRY(hipHostFree(s->pin_rec));
        s->pin_rec = nullptr;
        s->pin_rec_cap = 0;
        const size_t cap = std::max<size_t>(n_rec, 4096);
        HIP_TRY(hipHostMalloc((void **)&s->pin_rec, sizeof(double) * cap, hipHostMallocDefault));
        s->pin_rec_cap = cap;
    }
    if (n_rec) HIP_TRY(hipMemcpyAsync(s->pin_rec, s->rec_buf, sizeof(double) * n_rec, hipMemcpyDeviceToHost, s->stream));
    s->marks_done = true;
    return OCC_OK;
}

static int run_impl(occ_sampler *s, int64_t n_iter, int64_t burnin, double *out_alpha, double *out_beta, double *out_tau, bool snapshot)
{
    Ctx &c = s->ctx;
    const int C = c.C, p = c.p, q = c.q;
    const int64_t keep = n_iter - burnin;
    const size_t rw = (size_t)q + p + 1;
    const size_t need = (size_t)C * keep * rw;
    if (need > s->rec_cap) {  // grow the record buffer (its address lives in the device descriptor)
        WAIT_TRY(s->stream);
        if (s->rec_buf) HIP_TRY(hipFree(s->rec_buf));
        s->rec_buf = nullptr;
        s->rec_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->rec_buf, sizeof(double) * need));
        s->rec_cap = need;
    }
    if (c.rec != s->rec_buf) {
        c.rec = s->rec_buf;
        HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    }
    int rc = open_window(s, n_iter, burnin, keep, snapshot, true);
    if (rc) return rc;
    std::vector<ChainScalars> h;
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    s->marks_done = false;

    // calibration (no graph yet): a few eager sequences measure the Krylov launches a solve needs;
    // the count of the last, warm-started solve sizes the captured graph
    int64_t done_min = 0;
    const char *force = std::getenv("OCC_FORCE_KRYLOV_CAP");  // tests
    if (std::getenv("OCC_EAGER_ONLY")) {  // counter collection cannot follow graph launches: same kernels, eager
        for (int64_t i = 0; i < n_iter; ++i)
            if ((rc = eager_sequence(s))) return rc;
        done_min = n_iter;
    } else if (s->persistent && !s->flag_sync) {
        if (!s->head[0] && (rc = build_graph(s, 0))) return rc;  // the solve is one launch: nothing to calibrate
        if (s->need_prologue && (rc = launch_prologue(s))) return rc;
    } else if (s->flag_sync || s->rsr.m > 0) {
        if (s->need_prologue && (rc = launch_prologue(s))) return rc;
        // the solve is one launch: nothing to calibrate.  The captured pair of iterations starts with one
        // sequence parity: an odd number of stepped iterations since the capture is realigned by one more step.
        if (s->head[0] && s->parity != s->graph_parity) {
            if ((rc = eager_sequence(s))) return rc;
            done_min = 1;
        }
        if (!s->head[0] && n_iter > 1 && (rc = build_graph(s, 0))) return rc;
        if (n_iter == 1 && done_min == 0) {
            if ((rc = eager_sequence(s))) return rc;
            done_min = 1;
        }
    } else if (!s->head[0]) {
        const int64_t n_calib = std::min<int64_t>(n_iter, 3);
        s->calib_max = 0;
        for (int64_t i = 0; i < n_calib; ++i) {
            if (i == n_calib - 1) s->calib_max = 0;
            if ((rc = eager_sequence(s))) return rc;
        }
        done_min = n_calib;
        int cap = std::max(4, s->calib_max + 2);
        if (force) cap = std::max(1, std::atoi(force));
        if (done_min < n_iter && (rc = build_graph(s, cap))) return rc;
    } else if (s->need_prologue) {
        if ((rc = launch_prologue(s))) return rc;
    }
    // the first side chain waits for "the previous k_z_ob": everything enqueued so far
    if (done_min < n_iter && s->side_enabled && !s->flag_sync && s->rsr.m == 0) HIP_TRY(hipEventRecord(s->ev_z[s->parity ^ 1], s->stream));
    const int64_t seq_per_enqueue = (s->flag_sync || s->rsr.m > 0) ? GRAPH_SEQ : 1;

    // (iteration, launches spent on a carried solve) of every chain after the previous batch: every sequence moves every
    // unfinished chain -- by one iteration, or by the captured launches of a solve it carries on (Ctl::koff)
    std::vector<std::pair<uint32_t, uint32_t>> seen((size_t)C, {0xffffffffu, 0xffffffffu});
    while (done_min < n_iter) {
        // every sequence advances each unfinished chain by one iteration, or (rarely) carries its eta
        // solve into the next sequence; finished chains idle.  No host work inside a batch.
        const int64_t left = n_iter - done_min;
        int64_t batch = std::min<int64_t>(left, s->graph_launches < 128 ? 32 : 256);
        batch = (batch + seq_per_enqueue - 1) / seq_per_enqueue * seq_per_enqueue;  // a surplus sequence idles (it_stop)
        const auto hl0 = std::chrono::steady_clock::now();
        for (int64_t b = 0; b < batch; b += seq_per_enqueue)
            if ((rc = enqueue_sequence(s))) return rc;
        const auto hl1 = std::chrono::steady_clock::now();
        s->graph_launches += batch;
        if ((rc = finish_marks(s, done_min + batch >= n_iter, (size_t)C * keep * rw))) return rc;
        if ((rc = read_scalars(s, h))) return rc;
        if (std::getenv("OCC_VERBOSE")) {
            const auto hl2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[occ] host: %.1f us per sequence to enqueue, %.1f us per sequence until the batch finished\n",
                         std::chrono::duration<double, std::micro>(hl1 - hl0).count() / batch,
                         std::chrono::duration<double, std::micro>(hl2 - hl0).count() / batch);
        }
        if ((rc = check_device_errors(s, h))) return rc;
        done_min = n_iter;
        unsigned long long tot = 0, sq = 0, solves = 0;
        for (int ch = 0; ch < C; ++ch) {
            done_min = std::min<int64_t>(done_min, (int64_t)h[ch].ctl[s->parity].it - (int64_t)h[ch].it_base);
            tot += h[ch].krylov_total; sq += h[ch].krylov_sq_total; solves += h[ch].solves;
        }
        // A batch that leaves an unfinished chain exactly where it was -- same iteration, same carried launches, no error
        // word -- cannot happen while the chain's window is open on the device: the call ends with what it saw instead of
        // enqueuing idle sequences for ever (round 3 recorded a call that never returned; the text below is what
        // a recurrence reports).
        for (int ch = 0; ch < C; ++ch) {
            const Ctl &ctl = h[ch].ctl[s->parity];
            const bool unfinished = (int64_t)ctl.it - (int64_t)h[ch].it_base < n_iter;
            if (unfinished && seen[(size_t)ch].first == ctl.it && seen[(size_t)ch].second == ctl.koff) {
                std::string msg = "no progress: a batch of " + std::to_string((long long)batch) + " sequences left chain " + std::to_string(ch) +
                                  " where it was (the window of iterations is not open on the device?); per chain {it, koff | other parity | mid | it_base, it_stop, err}:";
                for (int k2 = 0; k2 < C && k2 < 16; ++k2) {
                    const ChainScalars &q2 = h[k2];
                    char b[192];
                    std::snprintf(b, sizeof(b), " [%d] {%u, %u | %u, %u | %u, %u / %u, %u | %u, %u, %d}", k2, q2.ctl[s->parity].it, q2.ctl[s->parity].koff, q2.ctl[s->parity ^ 1].it,
                                  q2.ctl[s->parity ^ 1].koff, q2.mid[0].it, q2.mid[0].koff, q2.mid[1].it, q2.mid[1].koff, q2.it_base, q2.it_stop, (int)q2.err);
                    msg += b;
                }
                msg += "; the host wanted " + std::to_string((long long)n_iter) + " iterations, done_min " + std::to_string((long long)done_min) + "; " + host_state(s);
                return set_error(s, OCC_E_HIP, msg.c_str());
            }
            seen[(size_t)ch] = {ctl.it, ctl.koff};
        }
        if (std::getenv("OCC_VERBOSE")) {
            unsigned long long car = 0;
            for (int ch = 0; ch < C; ++ch) car += h[ch].carries;
            std::fprintf(stderr, "[occ] batch of %lld sequences: done_min %lld / %lld, carries so far %llu, cap %d\n",
                         (long long)batch, (long long)done_min, (long long)n_iter, car, s->krylov_cap);
        }
        // re-size the captured solve from the solves since the last decision: mean + 2.5 sd
        const unsigned long long ds = solves - s->seen_solves;
        if (!force && !s->persistent && s->rsr.m == 0 && ds >= 32 && done_min < n_iter) {
            const double mean = (double)(tot - s->seen_tot) / ds;
            const double var = std::max(0.0, (double)(sq - s->seen_sq) / ds - mean * mean);
            const int want = std::max(4, (int)std::ceil(mean + 2.5 * std::sqrt(var)));
            s->seen_tot = tot; s->seen_sq = sq; s->seen_solves = solves;
            // hysteresis: grow at once, shrink only when two launches too many are captured
            if ((want > s->krylov_cap || want < s->krylov_cap - 1) && (rc = build_graph(s, want))) return rc;
        }
    }
    if (!s->marks_done) {  // (every iteration was stepped eagerly: nothing was enqueued above)
        if ((rc = finish_marks(s, true, (size_t)C * keep * rw))) return rc;
        if ((rc = read_scalars(s, h))) return rc;
        if ((rc = check_device_errors(s, h))) return rc;
    }
    // (ev1 and the records were enqueued behind the last batch; the read of the scalars that followed waited for both)
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->last_run_ms = ms;
    s->iterations = h[0].ctl[s->parity].it;
    s->krylov_last = h[0].minres_itn_last;
    s->mirror = h;  // the chains' scalars as this call leaves them: the next call does not ask the device again
    s->mirror_ok = true;
    s->clean_exit = true;

    const double *host = s->pin_rec;
    for (int ch = 0; ch < C; ++ch)
        for (int64_t t = 0; t < keep; ++t) {
            const double *row = host + ((size_t)ch * keep + t) * rw;
            std::copy(row, row + q, out_alpha + ((size_t)ch * keep + t) * q);
            std::copy(row + q, row + q + p, out_beta + ((size_t)ch * keep + t) * p);
            out_tau[(size_t)ch * keep + t] = row[q + p];
        }
    return OCC_OK;
}


// ---- run-time fallback of the fused iteration kernel ------------------------------------------------
// k_iter's barriers need every workgroup of a chain resident at once.  The residency probe at creation checks that
// on the device as it is then; another tenant of the device (a second engine, another process) can still take the CU
// slots later.  A barrier that gives up sets the chain's error word, the enqueued batch drains in milliseconds (every
// kernel skips such a chain), and the call is RE-RUN from the state it started with on the launch-per-step path: the
// variates are functions of (key, iteration, index) and both paths return the same bits, so the caller gets exactly
// what the fused path would have returned.  The engine stays on the launch-per-step path afterwards
// (occ_stats::persistent_solve = 0, ::fused_fallbacks counts).
// Hand-overs between the two streams: device counters (`flags`) or events / one stream.  The captured graphs belong to a
// mode and go with it; the counters restart from zero (a consistent state: sequence 0, whatever the parity).
static int set_handover(occ_sampler *s, bool flags)
{
    if (s->flag_sync == flags) return OCC_OK;
    WAIT_TRY(s->stream);
    WAIT_TRY(s->side);
    destroy_graph(s);
    s->flag_sync = flags;
    s->ctx.sync = flags ? s->sync_buf : nullptr;
    s->iter.sync = s->ctx.sync;
    s->rsr.sync = s->ctx.sync;
    if (flags) HIP_TRY(fill_on(s, s->sync_buf, 0, sizeof(unsigned) * SYNC_WORDS));
    HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    return OCC_OK;
}

static int fallback_to_launch_per_step(occ_sampler *s)
{
    Ctx &c = s->ctx;
    const size_t Cn = (size_t)c.C * c.n;
    WAIT_TRY(s->stream);
    WAIT_TRY(s->side);
    (void)hipGetLastError();
    if (std::getenv("OCC_VERBOSE") || !std::getenv("OCC_QUIET"))
        std::fprintf(stderr, "[occ] %s -- re-running the call without hand-overs between the streams%s\n",
                     s->err.c_str(), s->rsr.m > 0 ? "" : ", one launch per MINRES step");
    destroy_graph(s);
    s->persistent = false;
    s->xcd_local = false;
    s->tiles = false;
    s->device_timeout = false;
    s->launch_rc = OCC_OK;
    int rc;
    if ((rc = demote_streams(s, true))) return rc;
    s->flag_sync = false;
    c.sync = nullptr;
    s->iter.sync = nullptr;
    s->rsr.sync = nullptr;
    HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    HIP_TRY(copy_on(s, c.eta, s->snap_eta, sizeof(double) * Cn, hipMemcpyDeviceToDevice));
    HIP_TRY(copy_on(s, c.z, s->snap_z, Cn, hipMemcpyDeviceToDevice));
    HIP_TRY(copy_on(s, c.Xv, s->snap_x, sizeof(double2) * Cn, hipMemcpyDeviceToDevice));
    if (s->rsr.m > 0) HIP_TRY(copy_on(s, s->rsr.theta, s->snap_theta, sizeof(double) * (size_t)c.C * s->rsr.m, hipMemcpyDeviceToDevice));
    for (auto &sc : s->snap_sc) sc.err = 0;
    if ((rc = write_scalars(s, s->snap_sc))) return rc;
    s->parity = s->snap_parity;
    s->need_prologue = true;
    s->calib_max = 0;
    s->fused_fallbacks += 1;
    s->demoted = true;                       // ... until try_repromote finds the device as creation found it
    s->promote_wait = s->promote_backoff;
    return OCC_OK;
}

// Back to the paths creation chose, after a run-time fallback.  A device-side wait gives up because of what ELSE is on the
// device at that moment (another process's kernels on the CUs the fused kernel's barrier needs; more live hardware queues
// than slots): transient conditions.  At the next call -- then after 2, 4, ... 64 calls -- the engine takes its CU
// partition again, asks the stream probe and the residency probe what creation asked, and returns to the fused kernel and
// the device-side hand-overs when both pass (occ_stats::repromotions); otherwise it stays where it is.  The paths are
// bitwise equal and omega_b / the noise of the coming iteration are where the next kernel looks for them: nothing else to do.
static int try_repromote(occ_sampler *s)
{
    if (!s->demoted || !s->pref.valid || std::getenv("OCC_NO_REPROMOTE")) return OCC_OK;
    if (--s->promote_wait > 0) return OCC_OK;
    Ctx &c = s->ctx;
    int rc;
    auto stay = [&](const char *why) {
        if (std::getenv("OCC_VERBOSE")) std::fprintf(stderr, "[occ] not back on the fused path: %s\n", why);
        s->promote_backoff = std::min(64, s->promote_backoff * 2);
        s->promote_wait = s->promote_backoff;
        return OCC_OK;
    };
    WAIT_TRY(s->stream);
    WAIT_TRY(s->side);
    destroy_graph(s);
    if (s->pref.main_cus > 0) {  // the CU partition
        std::string why;
        StreamPair *pr = acquire_pair(s->device, s->pref.m_main, s->pref.m_side, &why);
        if (!pr) return stay(why.c_str());
        drop_pair(s);
        adopt_pair(s, pr);
        s->main_cus = s->pref.main_cus;
        c.share_on = s->pref.share_on;
    }
    auto back_out = [&](const char *why) -> int {  // to the demoted state
        s->persistent = false;
        s->xcd_local = false;
        s->tiles = false;
        int brc = demote_streams(s, true);
        if (brc) return brc;
        s->flag_sync = false;
        c.sync = nullptr; s->iter.sync = nullptr; s->rsr.sync = nullptr;
        HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
        return stay(why);
    };
    bool ok = true;
    if (s->pref.flag_sync) {
        if ((rc = stream_probe(s, &ok))) return rc;
        if (!ok) return back_out("the two streams do not run beside each other");
    }
    s->persistent = s->pref.persistent;
    s->xcd_local = s->pref.xcd_local;
    s->tiles = s->pref.tiles;
    HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    if (s->pref.persistent) {
        if ((rc = residency_probe(s, &ok))) return rc;
        if (!ok) return back_out("the fused kernel's workgroups are not resident together");
    }
    s->flag_sync = false;
    if ((rc = set_handover(s, s->pref.flag_sync))) return rc;
    if (!s->pref.flag_sync) HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    s->streams_serialised = false;
    s->demoted = false;
    s->promote_backoff = 1;
    s->repromotions += 1;
    if (std::getenv("OCC_VERBOSE")) std::fprintf(stderr, "[occ] back on the paths creation chose (fused kernel %d, device-side hand-overs %d)\n", (int)s->persistent, (int)s->flag_sync);
    return OCC_OK;
}

// Head of every occ_run / occ_step: come back from a fallback when the device allows it; ask the stream probe again
// when the process's set of streams has changed since the last answer (a queue can lose its hardware slot AFTER creation).
static int refresh_paths(occ_sampler *s)
{
    int rc;
    if ((rc = try_repromote(s))) return rc;
    // (a negative answer is not for life either: asked again after 1, 2, 4 ... 64 calls, as try_repromote does)
    bool ask = s->probe_gen != g_stream_gen.load();
    if (!ask && s->streams_serialised && !s->flag_sync && --s->probe_wait <= 0) ask = true;
    if (!s->demoted && s->main_cus > 0 && s->sync_buf && s->pref.flag_sync && ask) {
        bool beside = true;
        if ((rc = stream_probe(s, &beside))) return rc;
        s->streams_serialised = !beside;
        if (beside) s->probe_backoff = 1;
        else { s->probe_backoff = std::min(64, s->probe_backoff * 2); s->probe_wait = s->probe_backoff; }
        if ((rc = set_handover(s, beside))) return rc;
    }
    return OCC_OK;
}

int occ_step(occ_sampler *s)
{
    if (!s) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    if ((rc = refresh_paths(s))) return rc;
    const bool fused = (s->persistent && s->rsr.m == 0) || s->flag_sync;  // paths with device-side waits: re-run without them if one gives up
    s->device_timeout = false;
    rc = step_impl(s, fused);
    if (rc == OCC_E_HIP && fused && s->device_timeout) {
        if ((rc = fallback_to_launch_per_step(s))) return rc;
        rc = step_impl(s, false);
    }
    return rc;
}

int occ_run(occ_sampler *s, int64_t n_iter, int64_t burnin, double *out_alpha, double *out_beta, double *out_tau)
{
    if (!s) return OCC_E_BADARG;
    if (n_iter < 1 || burnin < 0 || burnin >= n_iter) return set_error(s, OCC_E_BADARG, "burnin value cannot be larger than sample size");
    if (!out_alpha || !out_beta || !out_tau) return set_error(s, OCC_E_BADARG, "null output buffer");
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    if ((rc = refresh_paths(s))) return rc;
    const bool fused = (s->persistent && s->rsr.m == 0) || s->flag_sync;
    s->device_timeout = false;
    rc = run_impl(s, n_iter, burnin, out_alpha, out_beta, out_tau, fused);
    if (rc == OCC_E_HIP && fused && s->device_timeout) {
        if ((rc = fallback_to_launch_per_step(s))) return rc;
        rc = run_impl(s, n_iter, burnin, out_alpha, out_beta, out_tau, false);
    }
    return rc;
}

int occ_get_state(occ_sampler *s, int32_t chain, const char *name, double *out, int64_t cap, int64_t *len)
{
    if (!s || !name || !len) return OCC_E_BADARG;
    const Ctx &c = s->ctx;
    if (chain < 0 || chain >= c.C) return set_error(s, OCC_E_BADARG, "bad chain index");
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    WAIT_TRY(s->stream);
    const std::string nm(name);
    const size_t n = (size_t)c.n, R = (size_t)c.R;
    std::vector<double> v;
    auto pull = [&](const double *src, size_t count) -> int {
        v.resize(count);
        if (count) HIP_TRY(copy_on(s, v.data(), src, sizeof(double) * count, hipMemcpyDeviceToHost));
        return OCC_OK;
    };
    int rc = OCC_OK;
    std::vector<ChainScalars> h;
    if ((rc = read_scalars(s, h))) return rc;
    const ChainScalars &sc = h[chain];
    const uint32_t it = sc.ctl[s->parity].it;  // iterations completed
    if (nm == "eta") rc = pull(c.eta + chain * n, n);
    else if (nm == "omega_b") rc = pull(c.omega_b[(it + 1) & 1] + chain * n, n);  // of the last completed iteration
    else if (nm == "omega_a") rc = pull(c.omega_a + chain * R, R);
    else if (nm == "rhs") rc = pull(c.rhs + chain * n, n);
    else if (nm == "z" || nm == "k" || nm == "exists") {
        std::vector<uint8_t> z(n);
        HIP_TRY(copy_on(s, z.data(), c.z + chain * n, n, hipMemcpyDeviceToHost));
        if (nm == "exists") {
            v.resize((size_t)c.S);
            for (int t = 0; t < c.S; ++t) v[t] = (s->obs_site[t] || z[s->site_id[t]]) ? 1.0 : 0.0;
        } else {
            v.resize(n);
            for (size_t i = 0; i < n; ++i) v[i] = (nm == "z") ? (double)z[i] : (double)z[i] - 0.5;
        }
    } else if (nm == "xz") {
        std::vector<double2> x(n);
        HIP_TRY(copy_on(s, x.data(), c.Xv + chain * n, sizeof(double2) * n, hipMemcpyDeviceToHost));
        v.resize(2 * n);
        for (size_t i = 0; i < n; ++i) { v[i] = x[i].x; v[n + i] = x[i].y; }
    } else if (nm == "alpha") v.assign(sc.alpha, sc.alpha + c.q);
    else if (nm == "beta") v.assign(sc.beta, sc.beta + c.p);
    else if (nm == "tau") v.assign(1, sc.tau);
    else if (nm == "minres_itn") v.assign(1, (double)sc.minres_itn_last);
    else if (nm == "iter") v.assign(1, (double)it);
    else if (nm == "rsr_gram" && s->rsr.m > 0) {  // K' diag(omega_b) K of the last theta update (upper tiles; tests)
        v.resize((size_t)s->rsr.m * s->rsr.m);
        HIP_TRY(copy_on(s, v.data(), s->rsr.gram + (size_t)chain * v.size(), sizeof(double) * v.size(), hipMemcpyDeviceToHost));
    }
    else if (nm == "theta" && s->rsr.m > 0) {
        v.resize((size_t)s->rsr.m);
        HIP_TRY(copy_on(s, v.data(), s->rsr.theta + (size_t)chain * s->rsr.m, sizeof(double) * v.size(), hipMemcpyDeviceToHost));
    }
    else return set_error(s, OCC_E_STATE, "unknown state name");
    if (rc) return rc;
    *len = (int64_t)v.size();
    if (out) {
        if (cap < (int64_t)v.size()) return set_error(s, OCC_E_STATE, "output buffer too small");
        std::copy(v.begin(), v.end(), out);
    }
    return OCC_OK;
}

int occ_set_state(occ_sampler *s, int32_t chain, const char *name, const double *in, int64_t len)
{
    if (!s || !name || !in) return OCC_E_BADARG;
    const Ctx &c = s->ctx;
    if (chain < 0 || chain >= c.C) return set_error(s, OCC_E_BADARG, "bad chain index");
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    WAIT_TRY(s->stream);
    const std::string nm(name);
    const size_t n = (size_t)c.n, R = (size_t)c.R;
    auto need = [&](size_t want) { return (size_t)len == want; };
    if (nm == "eta") {
        if (!need(n)) return set_error(s, OCC_E_STATE, "wrong length");
        HIP_TRY(copy_on(s, c.eta + chain * n, in, sizeof(double) * n, hipMemcpyHostToDevice));
    } else if (nm == "omega_a") {
        if (!need(R)) return set_error(s, OCC_E_STATE, "wrong length");
        HIP_TRY(copy_on(s, c.omega_a + chain * R, in, sizeof(double) * R, hipMemcpyHostToDevice));
    } else if (nm == "z") {
        if (!need(n)) return set_error(s, OCC_E_STATE, "wrong length");
        std::vector<uint8_t> z(n);
        for (size_t i = 0; i < n; ++i) z[i] = in[i] != 0.0;
        HIP_TRY(copy_on(s, c.z + chain * n, z.data(), n, hipMemcpyHostToDevice));
    } else if (nm == "theta" && s->rsr.m > 0) {
        if (!need((size_t)s->rsr.m)) return set_error(s, OCC_E_STATE, "wrong length");
        int rc = set_theta(s, chain, in);
        if (rc) return rc;
    } else if (nm == "xz") {
        if (!need(2 * n)) return set_error(s, OCC_E_STATE, "wrong length");
        std::vector<double2> x(n);
        for (size_t i = 0; i < n; ++i) x[i] = make_double2(in[i], in[n + i]);
        HIP_TRY(copy_on(s, c.Xv + chain * n, x.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    } else if (nm == "debug_close_window") {
        // tests of the no-progress exit of occ_run: the next call's window reaches the device with zero iterations
        if (!need(1)) return set_error(s, OCC_E_STATE, "wrong length");
        s->debug_close_window = in[0] != 0.0;
        return OCC_OK;
    } else if (nm == "debug_maxiter") {
        // tests of the MINRES-did-not-converge exit (logit.py:91-92): the iteration limit of every chain's solve; 0 restores
        // scipy's default 5 * (2n).  The limit travels by value in the kernels' argument blocks: the graphs are re-captured.
        if (!need(1) || in[0] < 0.0) return set_error(s, OCC_E_STATE, "wrong length");
        WAIT_TRY(s->stream);
        WAIT_TRY(s->side);
        destroy_graph(s);
        s->ctx.maxiter = in[0] > 0.0 ? (long long)in[0] : 10LL * c.n;
        s->kry.maxiter = s->ctx.maxiter;
        s->iter.a.maxiter = s->ctx.maxiter;
        HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    } else {
        std::vector<ChainScalars> h;
        int rc = read_scalars(s, h);
        if (rc) return rc;
        ChainScalars &sc = h[chain];
        if (nm == "alpha" && need((size_t)c.q)) std::copy(in, in + c.q, sc.alpha);
        else if (nm == "beta" && need((size_t)c.p)) std::copy(in, in + c.p, sc.beta);
        else if (nm == "tau" && need(1)) sc.tau = in[0];
        else if (nm == "iter" && need(1)) {
            const Ctl fresh = {(uint32_t)in[0], 0u};
            sc.ctl[0] = sc.ctl[1] = sc.mid[0] = sc.mid[1] = fresh;
        } else return set_error(s, OCC_E_STATE, "unknown state name or wrong length");
        if ((rc = write_scalars(s, h))) return rc;
    }
    s->need_prologue = true;  // omega_b of the coming iteration must be redrawn from the new state
    return OCC_OK;
}

int occ_get_stats(occ_sampler *s, occ_stats *out)
{
    if (!s || !out) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<ChainScalars> h;
    int rc = read_scalars(s, h);
    if (rc) return rc;
    unsigned long long tot = 0, solves = 0, carries = 0;
    for (auto &sc : h) { tot += sc.krylov_total; solves += sc.solves; carries += sc.carries; }
    s->stalls = (int64_t)carries;
    out->iterations = h[0].ctl[s->parity].it;
    out->graph_launches = s->graph_launches;
    out->eager_iterations = s->eager_iterations;
    out->stalls = s->stalls;
    out->krylov_cap = s->krylov_cap;
    out->krylov_last = h[0].minres_itn_last;
    out->krylov_mean = solves ? (double)tot / (double)solves : 0.0;
    out->krylov_total = (int64_t)tot;
    out->solves = (int64_t)solves;
    out->last_run_ms = s->last_run_ms;
    out->n_blocks_sites = s->ctx.nb_n;
    out->n_blocks_rows = s->ctx.nb_r;
    out->threads_per_block = s->tpb;
    out->n_chains = s->ctx.C;
    out->persistent_solve = s->persistent ? (s->tiles ? 3 : s->xcd_local ? 2 : 1) : 0;
    out->solve_workgroups = s->tiles ? s->tiles_G : s->iter.nbg;
    out->main_stream_cus = s->main_cus;
    out->fused_fallbacks = (int32_t)s->fused_fallbacks;
    out->repromotions = (int32_t)s->repromotions;
    out->stream_probes = (int32_t)s->stream_probes;
    out->handover_mode = s->flag_sync ? 2 : 1;
    {
        int masked = 0, plain = 0, idle = 0;
        count_pairs(s->device, &masked, &plain, &idle);
        out->stream_pairs_masked = masked;
        out->stream_pairs_plain = plain;
        out->stream_pairs_idle = idle;
        out->stream_pairs_evicted = (int32_t)g_pairs_evicted.load();
    }
    out->demoted = s->demoted ? 1 : 0;
    out->profile_iter_dispatch_us = s->profile_iter_dispatch_us;
    out->profile_minres_iterations = s->profile_minres_iterations;
    out->iter_kernel_launches = 0;
    out->iter_kernel_mean_us = 0.0;
    if (s->ctx.iter_clock) {
        unsigned long long clk[4];
        HIP_TRY(copy_on(s, clk, s->ctx.iter_clock, sizeof(clk), hipMemcpyDeviceToHost));
        int khz = 0;
        HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device));
        out->iter_kernel_launches = (int64_t)clk[3];
        if (clk[3] && khz > 0) out->iter_kernel_mean_us = 1000.0 * (double)clk[2] / (double)khz / (double)clk[3];
    }
    return OCC_OK;
}

// Average launch-to-launch time of one kernel kind inside a hipGraph (the mode occ_run uses): `reps`
// back-to-back launches of ONE kernel are captured into a graph and bracketed by two HIP events on
// the engine's stream; elapsed / reps = kernel duration + the dependent-launch boundary.  Eager
// launches are not representative (idle gaps, end-of-kernel flushes before host copies).
static int time_kernel_graph(occ_sampler *s, int kind, int reps, int e, int extra, double *avg_us)
{
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < reps; ++r) LAUNCH(s, s->stream, kind, e, extra);
    HIP_TRY(hipStreamEndCapture(s->stream, &graph));
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(exec, s->stream));  // untimed: instruction cache, clocks
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    HIP_TRY(hipGraphLaunch(exec, s->stream));
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    WAIT_TRY(s->stream);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *avg_us = 1000.0 * ms / reps;
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    return take_launch_rc(s);
}

int occ_profile(occ_sampler *s, int32_t reps, int64_t counts[OCC_N_KERNEL_KINDS], double total_us[OCC_N_KERNEL_KINDS])
{
    if (!s || reps < 1 || !counts || !total_us) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    HIP_TRY(hipSetDevice(s->device));
    int rc = open_window(s, 1 << 30, 0, 0, false, false);  // no chain reaches its stop during the timing loops
    if (rc) return rc;
    // The fused iteration kernel first, IN SITU: `reps` real iterations continue the chains from where they are
    // (nothing recorded), the side chain on the side stream as in occ_run, two HIP events around every
    // k_iter launch on the main stream.  counts[K_ITER] = launches, total_us[K_ITER] = the sum of their durations;
    // the MINRES iterations they ran show in occ_get_stats (krylov_total, solves) before and after.
    counts[K_ITER] = 0;
    total_us[K_ITER] = 0.0;
    if (s->persistent) {
        std::vector<ChainScalars> h0;
        if ((rc = read_scalars(s, h0))) return rc;
        if (s->need_prologue && (rc = launch_prologue(s))) return rc;
        // hand-overs as in occ_run: device counters, or stream events; OCC_EAGER_ONLY (counter collection
        // serialises kernels): everything on the main stream in the reference's order, no hand-overs at all
        const bool one_stream = std::getenv("OCC_EAGER_ONLY") != nullptr || !s->side_enabled;
        const bool flags = s->flag_sync && !one_stream;
        const bool ev = !flags && !one_stream;
        const bool old_sync = s->launch_sync;
        s->launch_sync = flags;
        if (ev) HIP_TRY(hipEventRecord(s->ev_z[s->parity ^ 1], s->stream));
        hipEvent_t pev[2] = {nullptr, nullptr};
        HIP_TRY(hipEventCreate(&pev[0]));
        HIP_TRY(hipEventCreate(&pev[1]));
        double dispatch_us = 0.0;
        int dispatch_n = 0;
        for (int r = 0; r < reps; ++r) {
            const int pe = s->parity;
            hipStream_t side = one_stream ? s->stream : s->side;
            if (ev) HIP_TRY(hipStreamWaitEvent(side, s->ev_z[pe ^ 1], 0));
            if (flags) LAUNCH(s, side, K_GATE, 0);
            LAUNCH(s, side, K_OMEGA_A, pe);
            LAUNCH(s, side, K_NOISE, pe, 1);
            if (ev) HIP_TRY(hipEventRecord(s->ev_side[pe], side));
            HIP_TRY(hipEventRecord(s->ev0, s->stream));
            s->ext_ev0 = pev[0];  // the dispatch's own start / stop (hipExtLaunchKernel): rocprofv3's basis for a kernel's duration
            s->ext_ev1 = pev[1];
            LAUNCH(s, s->stream, K_ITER, pe);
            s->ext_ev0 = s->ext_ev1 = nullptr;
            HIP_TRY(hipEventRecord(s->ev1, s->stream));
            if (ev) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_side[pe], 0));
            LAUNCH(s, s->stream, K_Z_OB, pe);
            if (ev) HIP_TRY(hipEventRecord(s->ev_z[pe], s->stream));
            s->parity ^= 1;
            WAIT_TRY(s->stream);
            // (the side stream too: launched one by one the main stream runs ahead of it, and the NEXT k_iter would spend the
            // head of its dispatch -- inside the start / stop events, outside its own clock -- waiting for this sequence's noise)
            if (!one_stream) WAIT_TRY(s->side);
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
            total_us[K_ITER] += 1000.0 * ms;
            if (hipEventElapsedTime(&ms, pev[0], pev[1]) == hipSuccess) { dispatch_us += 1000.0 * ms; dispatch_n += 1; }
            else (void)hipGetLastError();
        }
        s->profile_iter_dispatch_us = dispatch_n ? dispatch_us / dispatch_n : 0.0;
        (void)hipEventDestroy(pev[0]);
        (void)hipEventDestroy(pev[1]);
        s->launch_sync = old_sync;
        counts[K_ITER] = reps;
        std::vector<ChainScalars> h;
        if ((rc = read_scalars(s, h))) return rc;
        if ((rc = check_device_errors(s, h))) return rc;
        unsigned long long dt = 0, ds = 0;
        for (size_t ch = 0; ch < h.size(); ++ch) { dt += h[ch].krylov_total - h0[ch].krylov_total; ds += h[ch].solves - h0[ch].solves; }
        s->profile_minres_iterations = ds ? (double)dt / (double)ds : 0.0;
    }
    const int e = s->parity;
    double us = 0.0;
    // the kernels below are replayed out of sequence: no waits, no counter updates
    struct NoSync { occ_sampler *s; bool old; explicit NoSync(occ_sampler *p) : s(p), old(p->launch_sync) { s->launch_sync = false; } ~NoSync() { s->launch_sync = old; } } no_sync(s);
    auto timed = [&](int kind, int extra) -> int {
        int r = time_kernel_graph(s, kind, reps, e, extra, &us);
        counts[kind] = reps;
        total_us[kind] = us * reps;
        return r;
    };
    // kernels that only read the state of the last completed iteration
    if ((rc = timed(K_OMEGA_B, 0))) return rc;
    if ((rc = timed(K_NOISE, 0))) return rc;
    if ((rc = timed(K_OMEGA_A, 0))) return rc;
    if ((rc = timed(K_ALPHA_DRAW, 0))) return rc;
    if ((rc = timed(K_ETA_INIT, 0))) return rc;
    // k_minres is timed inside the launch sequence it really runs in: a graph of k_eta_init followed by
    // launches 1..KRY_TIMED of one solve (all of them do full work: no solve of these systems stops
    // that early), replayed `reps` times; the separately timed k_eta_init is subtracted.  Unlike a loop
    // over one repeated launch, every launch then reads what the previous one wrote (cache-cold, the
    // state of the real solve).
    {
        constexpr int KRY_TIMED = 8;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        LAUNCH(s, s->stream, K_ETA_INIT, e);
        for (int k = 1; k <= KRY_TIMED; ++k) LAUNCH(s, s->stream, K_MINRES, e, k);
        HIP_TRY(hipStreamEndCapture(s->stream, &graph));
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphLaunch(exec, s->stream));
        HIP_TRY(hipEventRecord(s->ev0, s->stream));
        for (int r = 0; r < reps; ++r) HIP_TRY(hipGraphLaunch(exec, s->stream));
        HIP_TRY(hipEventRecord(s->ev1, s->stream));
        WAIT_TRY(s->stream);
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        const double per_replay_us = 1000.0 * ms / reps;
        const double eta_us = total_us[K_ETA_INIT] / counts[K_ETA_INIT];
        counts[K_MINRES] = (int64_t)reps * KRY_TIMED;
        total_us[K_MINRES] = std::max(0.0, per_replay_us - eta_us) * reps;
    }
    // finish that solve so that the tail kernels have real work
    int k_last = 0;
    LAUNCH(s, s->stream, K_ETA_INIT, e);
    if ((rc = eager_krylov(s, 1, &k_last))) return rc;
    if ((rc = timed(K_BETA_PARTIAL, k_last))) return rc;
    // k_z_ob advances the control word of the OTHER parity; launched repeatedly with the same parity
    // it redoes the same z update and omega_b draw
    if ((rc = timed(K_Z_OB, 0))) return rc;
    WAIT_TRY(s->stream);
    s->need_prologue = true;
    return take_launch_rc(s);
}

// ---- per-conditional entry points with injected variates ---------------------------------
// One conditional of ONE chain with the kernels of the launch-per-step path (grid over that chain only), the variates
// coming from the caller.  cond_begin gives the chain a one-iteration window at its current iteration number and
// returns that number; the control words are not advanced by any of the entry points.
namespace {

#define OCC_CARGS s->ctx_dev, s->ctx.sc, s->ctx.slots, chain, e

int cond_begin(occ_sampler *s, int chain, const Inject &inj, uint32_t *it_out)
{
    Ctx &c = s->ctx;
    if (chain < 0 || chain >= c.C) return set_error(s, OCC_E_BADARG, "bad chain index");
    if (s->rsr.m > 0) return set_error(s, OCC_E_BADARG, "the per-conditional entry points cover the ICAR model");
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    if (!s->inj_dev) {
        if ((rc = dev_alloc(s, &s->inj_dev, 1))) return rc;
        if ((rc = dev_alloc(s, &s->inj_u, (size_t)c.n))) return rc;
        c.inj = s->inj_dev;
        HIP_TRY(copy_on(s, s->ctx_dev, &s->ctx, sizeof(Ctx), hipMemcpyHostToDevice));
    }
    std::vector<ChainScalars> h;
    if ((rc = read_scalars(s, h))) return rc;  // (synchronises both streams)
    ChainScalars &sc = h[chain];
    Ctl &ctl = sc.ctl[s->parity];
    ctl.koff = 0;
    sc.it_base = ctl.it;
    sc.it_stop = ctl.it + 1u;
    sc.burnin = 0; sc.keep = 0; sc.err = 0;
    *it_out = ctl.it;
    if ((rc = write_scalars(s, h))) return rc;
    Inject hinj = inj;
    hinj.z_u = s->inj_u;
    HIP_TRY(copy_on(s, s->inj_dev, &hinj, sizeof(Inject), hipMemcpyHostToDevice));
    s->need_prologue = true;  // whatever follows, omega_b of the coming iteration must be redrawn from the state
    return OCC_OK;
}

int cond_end(occ_sampler *s, int chain, ChainScalars *out)
{
    WAIT_TRY(s->stream);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { s->err = std::string("kernel launch failed: ") + hipGetErrorString(le); return OCC_E_HIP; }
    std::vector<ChainScalars> h;
    int rc = read_scalars(s, h);
    if (rc) return rc;
    if ((rc = check_device_errors(s, h))) return rc;
    if (out) *out = h[chain];
    return OCC_OK;
}

int copy_in(occ_sampler *s, double *dst, const double *src, size_t count)
{
    if (!src) return set_error(s, OCC_E_BADARG, "null input pointer");
    HIP_TRY(copy_on(s, dst, src, sizeof(double) * count, hipMemcpyDefault));
    return OCC_OK;
}
int copy_out(occ_sampler *s, double *dst, const double *src, size_t count)
{
    if (dst) HIP_TRY(copy_on(s, dst, src, sizeof(double) * count, hipMemcpyDeviceToHost));
    return OCC_OK;
}

}  // namespace

int occ_cond_tau(occ_sampler *s, int32_t chain, double gamma_variate, double *tau_out)
{
    if (!s) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    Inject inj{};
    inj.gamma = gamma_variate;
    inj.tau_from_gamma = 1;
    uint32_t it;
    int rc = cond_begin(s, chain, inj, &it);
    if (rc) return rc;
    const Ctx &c = s->ctx;
    const int e = s->parity;
    const dim3 blk((unsigned)s->tpb), gs((unsigned)c.nb_n, 1u);
    hipLaunchKernelGGL(k_quad, gs, blk, 0, s->stream, OCC_CARGS);
    hipLaunchKernelGGL(k_eta_init<1>, gs, blk, 0, s->stream, OCC_CARGS);  // tau = (1 / rate) gamma; its right-hand side is not used
    ChainScalars sc;
    if ((rc = cond_end(s, chain, &sc))) return rc;
    if (tau_out) *tau_out = sc.tau;
    return OCC_OK;
}

int occ_cond_eta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps_site, const double *prior_term, double *rhs_out,
                 double *xz_out, double *eta_out, int32_t *itn_out)
{
    if (!s) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    Inject inj{};
    inj.tau_from_gamma = 0;  // tau is the chain's
    uint32_t it;
    int rc = cond_begin(s, chain, inj, &it);
    if (rc) return rc;
    const Ctx &c = s->ctx;
    const size_t n = (size_t)c.n, co = (size_t)chain * n;
    if ((rc = copy_in(s, c.omega_b[it & 1] + co, omega_b, n))) return rc;
    if ((rc = copy_in(s, c.enorm[it & 1] + co, eps_site, n))) return rc;
    if ((rc = copy_in(s, c.uprior[it & 1] + co, prior_term, n))) return rc;
    const int e = s->parity;
    const dim3 blk((unsigned)s->tpb), gs((unsigned)c.nb_n, 1u);
    hipLaunchKernelGGL(k_eta_init<1>, gs, blk, 0, s->stream, OCC_CARGS);
    int k_last = 0;
    Slot slot;
    // OCC_DEBUG_EXACT_DIV=1: the scalar recurrence in scipy's own arithmetic (k_minres<1>, occ_kernels.hpp
    // minres_scalars_exact) -- the tight-tolerance check of the vector part against the reference's recorded solves
    const char *xd = std::getenv("OCC_DEBUG_EXACT_DIV");
    const bool exact_div = xd && std::atoi(xd) != 0;
    for (int k = 1;; ++k) {  // one launch per MINRES step, the host watching this chain's `done` flag
        if (exact_div) hipLaunchKernelGGL(k_minres<1>, gs, blk, 0, s->stream, s->kry, chain, e, k);
        else hipLaunchKernelGGL(k_minres<0>, gs, blk, 0, s->stream, s->kry, chain, e, k);
        if (k < 4) continue;
        HIP_TRY(hipMemcpyAsync(&slot, c.slots + (size_t)chain * NSLOT + (k & (NSLOT - 1)), sizeof(Slot), hipMemcpyDeviceToHost, s->stream));
        WAIT_TRY(s->stream);
        if (slot.done) { k_last = k; break; }
        if ((long long)k > c.maxiter + 3) return set_error(s, OCC_E_MINRES, "MINRES solver did not converge!");
    }
    hipLaunchKernelGGL(pick_beta_partial(s->generic ? 0 : c.p), gs, blk, s->generic ? generic_lds_bytes(nacc(c.p), s->tpb) : 0, s->stream, OCC_CARGS, k_last);
    ChainScalars sc;
    if ((rc = cond_end(s, chain, &sc))) return rc;
    if ((rc = copy_out(s, rhs_out, c.rhs + co, n))) return rc;
    if ((rc = copy_out(s, eta_out, c.eta + co, n))) return rc;
    if (xz_out) {
        std::vector<double2> x(n);
        HIP_TRY(copy_on(s, x.data(), c.Xv + co, sizeof(double2) * n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) { xz_out[i] = x[i].x; xz_out[n + i] = x[i].y; }
    }
    if (itn_out) *itn_out = sc.minres_itn_last;
    return OCC_OK;
}

int occ_cond_beta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps, double *beta_out)
{
    if (!s || !eps) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    Inject inj{};
    inj.do_beta = 1;
    HIP_TRY(copy_on(s, inj.beta_eps, eps, sizeof(double) * s->ctx.p, hipMemcpyDefault));
    uint32_t it;
    int rc = cond_begin(s, chain, inj, &it);
    if (rc) return rc;
    const Ctx &c = s->ctx;
    const size_t n = (size_t)c.n, co = (size_t)chain * n;
    if ((rc = copy_in(s, c.omega_b[it & 1] + co, omega_b, n))) return rc;
    const int e = s->parity;
    const dim3 blk((unsigned)s->tpb), gs((unsigned)c.nb_n, 1u);
    if (s->generic) {
        hipLaunchKernelGGL(k_beta_sums<0>, gs, blk, generic_lds_bytes(nacc(c.p), s->tpb), s->stream, OCC_CARGS);
        hipLaunchKernelGGL((k_beta_draw<0, 1>), dim3(1), dim3(256), 0, s->stream, OCC_CARGS);
    } else {
        hipLaunchKernelGGL(OCC_PICK_P(k_beta_sums, c.p), gs, blk, 0, s->stream, OCC_CARGS);
        hipLaunchKernelGGL(OCC_PICK_P(k_cond_beta_z, c.p), gs, blk, 0, s->stream, OCC_CARGS);
    }
    ChainScalars sc;
    if ((rc = cond_end(s, chain, &sc))) return rc;
    if (beta_out) std::copy(sc.beta, sc.beta + c.p, beta_out);
    return OCC_OK;
}

int occ_cond_alpha(occ_sampler *s, int32_t chain, const double *omega_a, const double *eps, double *alpha_out)
{
    if (!s || !eps) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    Inject inj{};
    HIP_TRY(copy_on(s, inj.alpha_eps, eps, sizeof(double) * s->ctx.q, hipMemcpyDefault));
    uint32_t it;
    int rc = cond_begin(s, chain, inj, &it);
    if (rc) return rc;
    const Ctx &c = s->ctx;
    if ((rc = copy_in(s, c.omega_a + (size_t)chain * c.R, omega_a, (size_t)c.R))) return rc;
    const int e = s->parity;
    const dim3 blk((unsigned)s->tpb), gr((unsigned)c.nb_r, 1u);
#define OCC_OMEGA_A_INJ(q) ((q) == 1 ? k_omega_a<1, 1> : (q) == 2 ? k_omega_a<2, 1> : (q) == 3 ? k_omega_a<3, 1> : (q) == 4 ? k_omega_a<4, 1> : (q) == 5 ? k_omega_a<5, 1> : (q) == 6 ? k_omega_a<6, 1> : (q) == 7 ? k_omega_a<7, 1> : k_omega_a<8, 1>)
    if (s->generic) hipLaunchKernelGGL((k_omega_a<0, 1>), gr, blk, generic_lds_bytes(nacc(c.q), s->tpb), s->stream, OCC_CARGS, 0);
    else hipLaunchKernelGGL(OCC_OMEGA_A_INJ(c.q), gr, blk, 0, s->stream, OCC_CARGS, 0);
    hipLaunchKernelGGL(k_alpha_draw<1>, dim3(1), dim3(512), 0, s->stream, OCC_CARGS, 0);
    ChainScalars sc;
    if ((rc = cond_end(s, chain, &sc))) return rc;
    if (alpha_out) std::copy(sc.alpha, sc.alpha + c.q, alpha_out);
    return OCC_OK;
}

int occ_cond_z(occ_sampler *s, int32_t chain, const double *u, double *z_out)
{
    if (!s) return OCC_E_BADARG;
    DeviceLease lease = lease_device(s->device);
    Inject inj{};
    inj.do_z = 1;
    uint32_t it;
    int rc = cond_begin(s, chain, inj, &it);
    if (rc) return rc;
    const Ctx &c = s->ctx;
    const size_t n = (size_t)c.n;
    if ((rc = copy_in(s, s->inj_u, u, n))) return rc;
    const int e = s->parity;
    const dim3 blk((unsigned)s->tpb), gs((unsigned)c.nb_n, 1u);
    hipLaunchKernelGGL(OCC_PICK_P(k_cond_beta_z, s->generic ? 0 : c.p), gs, blk, 0, s->stream, OCC_CARGS);
    if ((rc = cond_end(s, chain, nullptr))) return rc;
    if (z_out) {
        std::vector<uint8_t> z(n);
        HIP_TRY(copy_on(s, z.data(), c.z + (size_t)chain * n, n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) z_out[i] = (double)z[i];
    }
    return OCC_OK;
}

// Device draws of the engine's own variate generators (see the header).
int occ_draw(int32_t device, int32_t kind, uint64_t key, uint32_t iteration, uint32_t stream, int64_t n, const double *param, double *out)
{
    if (n < 0 || n > 0x7fffffffLL || !out || kind < 0 || kind > 4 || ((kind == 0 || kind == 1 || kind == 4) && !param && n > 0) || (kind == 4 && n % 64 != 0)) {
        g_create_error = "occ_draw: bad arguments";
        return OCC_E_BADARG;
    }
    if (n == 0) return OCC_OK;
    DeviceLease lease = lease_device(device);  // (a null-stream launch synchronises with the blocking streams of the device)
    double *d_par = nullptr, *d_out = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(double) * (size_t)n);
    if (e == hipSuccess && param) {
        e = hipMalloc((void **)&d_par, sizeof(double) * (size_t)n);
        if (e == hipSuccess) e = hipMemcpy(d_par, param, sizeof(double) * (size_t)n, hipMemcpyDefault);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (int)kind, key, iteration, stream, (long long)n,
                           (const double *)d_par, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
    if (d_par) (void)hipFree(d_par);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) {
        g_create_error = std::string("occ_draw: ") + hipGetErrorString(e);
        return OCC_E_HIP;
    }
    return OCC_OK;
}

#ifdef OCC_SOLVE_STAMPS
// developer build only: the time stamps of the last k_iter launch (chain 0, workgroup 0)
int occ_debug_solve_stamps(unsigned long long *out, int capacity)
{
    const int n = STAMP_STEPS * STAMP_POINTS;
    if (capacity < n) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_solve_stamps), sizeof(unsigned long long) * n) != hipSuccess) return -2;
    return n;
}
#endif

}  // extern "C"
