/*
 * occ_gibbs.h -- C ABI of the MI355X (gfx950) Gibbs engine for the ICAR spatial occupancy model.
 *
 * The reference (zoj613/OccuSpytial v0.2.0) has no FFI boundary of its own for this path: the hot
 * loop is the Python method LogitICARGibbs.step() (occuspytial/gibbs/logit.py:254-266) calling
 * numpy/scipy, two Cython helpers and the third-party `polyagamma` C sampler.  This header is the
 * boundary a binding of that path would use: one opaque handle per (device, batch of chains), plain
 * pointers and sizes, int status codes that map onto the reference's exception types/messages.
 * The Python classes occuspytial_amd.gibbs.LogitICARGibbs / LogitRSRGibbs bind it with ctypes
 * (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *  - every function returns OCC_OK (0) or a negative OCC_E* code; occ_last_error() gives the text;
 *  - input pointers may be host or device memory (copied with hipMemcpyDefault); the library never
 *    frees or keeps caller memory; outputs are host buffers owned by the caller;
 *  - all real data is IEEE float64, C-contiguous; index arrays are int32;
 *  - a handle is driven by one host thread at a time; handles on DIFFERENT devices are independent, and ctypes
 *    releases the GIL so host threads can drive several GPUs (occ_create_group;
 *    occuspytial_amd.gibbs.LogitICARGibbs(..., devices=[...]) does exactly that); handles on the SAME device share
 *    that device's pooled streams and their calls run one after the other (a per-device lock inside the library);
 *  - there is NO CPU fallback: creation fails with OCC_E_HIP when no gfx950 device is usable.
 */
#ifndef OCC_GIBBS_H
#define OCC_GIBBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCC_ABI_VERSION 6
#define OCC_MAX_COVARIATES 32 /* p and q limit.  Up to 8 of each: kernels with the p x p / q x q accumulators in registers and the
                                fused iteration kernel; 9 to 32: generic kernels (run-time p and q, terms reduced one at a time,
                                Cholesky factor in LDS) on the launch-per-step path */

enum {
    OCC_OK = 0,
    OCC_E_BADARG = -1,   /* -> ValueError */
    OCC_E_HIP = -2,      /* -> RuntimeError (HIP runtime failure / no device) */
    OCC_E_MINRES = -3,   /* -> RuntimeError('MINRES solver did not converge!')        logit.py:91-92 */
    OCC_E_CHOLESKY = -4, /* -> RuntimeError('Cholesky factorization/solver failed!')  distributions.pyx:21,107-108 */
    OCC_E_STATE = -5     /* unknown state name / wrong length */
};

/* The fixed inputs of LogitICARGibbs(Q, W, X, y, hparams) (logit.py:172-174, base.py:84-88,107-162)
 * flattened by the host: W/y dictionaries become one (R x q) matrix / R vector in surveyed-site
 * order with site_ptr offsets (replaces the Data container, data.pyx:61-140); Q is CSR with sorted
 * columns (the reference keeps CSC of the same symmetric matrix, base.py:122). */
typedef struct occ_problem {
    int64_t n;               /* sites                                   base.py:123 */
    int64_t n_surveyed;      /* S = len(W)                              data.pyx:142-144 */
    int64_t n_rows;          /* R = total visits over surveyed sites */
    int32_t p;               /* occupancy covariates  (X.shape[1]) */
    int32_t q;               /* detection covariates  (W[i].shape[1]) */
    const int32_t *q_indptr; /* n+1 */
    const int32_t *q_indices;
    const double *q_data;    /* ICAR precision: zero row sums, non-positive off-diagonals */
    const double *X;         /* n x p, row-major */
    const int32_t *site_id;  /* S: site number of each surveyed site (Data.surveyed order) */
    const int32_t *site_ptr; /* S+1: row offsets into W / y */
    const double *W;         /* R x q, row-major */
    const double *y;         /* R: 0/1 detections */
    const double *a_mu, *a_prec; /* q, q x q    base.py:49-61,177-186 */
    const double *b_mu, *b_prec; /* p, p x p */
    double tau_rate, tau_shape;
    /* Reduced-rank model (LogitRSRGibbs, logit.py:269-485), optional: rsr_dim = m > 0 selects it.  The spatial
     * effects are eta = K theta with K the n x m Moran-operator basis the host computed (logit.py:413-446);
     * rsr_Q = K'QK (m x m), rsr_E its eigenfactor, E E' = K'QK (logit.py:321-323); all row-major; m <= 4096 (up to 128 the
     * system is solved in LDS and registers, beyond that panel by panel in device memory). */
    int32_t rsr_dim;
    const double *rsr_K, *rsr_Q, *rsr_E;
    /* Prior draw of the eta conditional, optional.  NULL (default): the engine draws the N(0, Q) prior term in EDGE form,
     * u = B'eps with Q = B'B the weighted incidence factorisation -- O(nnz) work, no set-up -- which needs Q to be an ICAR
     * precision D - W with W >= 0.  Non-NULL: the reference's own form (logit.py:64-67, 77): u = F eps with F an
     * n x prior_factor_cols matrix, row-major, F F' = Q (the reference's eigenfactor E = U[:, 1:] sqrt(s[1:]) from the
     * dense eigh of Q; any factor gives the same law) and eps standard normals of Philox stream 10 -- O(n^2) memory and
     * bytes per iteration, but ANY symmetric positive semi-definite singular Q is accepted (positive off-diagonals too). */
    const double *prior_factor;
    int64_t prior_factor_cols;
} occ_problem;

typedef struct occ_sampler occ_sampler;

/* Build device-resident state for `n_chains` independent chains of one problem on HIP device
 * `device`.  keys[c] seeds chain c's counter-based (Philox4x32-10) variate streams; the host derives
 * it from the chain's numpy SeedSequence (base.py:88, 293-306).  Replaces GibbsBase.__init__ /
 * _configure (base.py:84-164) and _EtaICARPosterior.__init__ (logit.py:64-71; no dense eigenfactor). */
int occ_create(const occ_problem *problem, int32_t n_chains, const uint64_t *keys, int32_t device,
               occ_sampler **out);
int occ_destroy(occ_sampler *s);
/* Replace the chains' Philox keys (a new sample() call on an existing sampler draws new keys from
 * its generator, exactly as the reference's rng stream simply continues, base.py:88). */
int occ_set_keys(occ_sampler *s, const uint64_t *keys);

/* Starting values of one chain: base.py:188-197 (`start` dict) / 199-212 (default start, drawn by
 * the host with numpy exactly as the reference does).  Resets the chain's iteration counter and the
 * MINRES warm start (logit.py:71).  Reduced-rank model: `eta` points at the m coefficients theta (logit.py:457-460). */
int occ_set_start(occ_sampler *s, int32_t chain, const double *alpha, const double *beta, double tau,
                  const double *eta);

/* One Gibbs iteration of every chain, launched kernel by kernel on one stream in the reference's order:
 * LogitICARGibbs.step() (logit.py:254-266). */
int occ_step(occ_sampler *s);

/* n_iter iterations of every chain; alpha/beta/tau of iterations >= burnin are recorded:
 * GibbsBase._run's loop (base.py:236-239) for all chains at once (gibbs/parallel.py:38-41).
 * out_alpha: [n_chains][n_iter-burnin][q], out_beta: [..][p], out_tau: [n_chains][n_iter-burnin].
 * The iteration is replayed from captured hipGraphs on two streams (omega_a/alpha overlap the eta solve,
 * see DESIGN.md). */
int occ_run(occ_sampler *s, int64_t n_iter, int64_t burnin, double *out_alpha, double *out_beta,
            double *out_tau);

/* State of one chain by name (reference attribute names, base.py:65-82 / logit.py):
 *   alpha(q) beta(p) tau(1) eta(n) z(n) k(n) omega_b(n) omega_a(R) exists(S) xz(2n) rhs(n)
 *   minres_itn(1) iter(1); reduced-rank model: theta(m), and eta is K theta (the reference's `spatial`)
 * occ_get_state copies into out (capacity cap doubles) and stores the length in *len.
 * occ_set_state accepts alpha beta tau eta z omega_a xz iter theta (theta also sets eta = K theta); omega_b of
 * the coming iteration is then redrawn from the new state. */
int occ_get_state(occ_sampler *s, int32_t chain, const char *name, double *out, int64_t cap, int64_t *len);
int occ_set_state(occ_sampler *s, int32_t chain, const char *name, const double *in, int64_t len);

typedef struct occ_stats {
    int64_t iterations;      /* Gibbs iterations completed (chain 0) */
    int64_t graph_launches;  /* hipGraph replays */
    int64_t eager_iterations;
    int64_t stalls;          /* eta solves carried into a second graph replay (more Krylov steps than captured) */
    int32_t krylov_cap;      /* Krylov steps captured per eta solve */
    int32_t krylov_last;     /* MINRES iterations of the last eta solve (chain 0) */
    double krylov_mean;      /* mean MINRES iterations per solve since creation (all chains) */
    int64_t krylov_total;    /* MINRES iterations since creation (all chains) */
    int64_t solves;          /* eta solves since creation (all chains) */
    double last_run_ms;      /* device time of the last occ_run (HIP events on the engine's stream) */
    int32_t n_blocks_sites, n_blocks_rows, threads_per_block, n_chains;
    int32_t persistent_solve; /* 1, 2: fused iteration kernel with the persistent eta solve (k_iter + k_z_ob per iteration;
                                 2: one XCD per chain, its workgroups exchange through that XCD's L2), 0: one launch per
                                 MINRES step, omega_a/alpha/noise on a side stream */
    int32_t solve_workgroups; /* workgroups per chain of the persistent solve */
    int32_t main_stream_cus;  /* > 0: CUs reserved for the main stream (k_iter, k_z_ob); the side stream has the others */
    int32_t fused_fallbacks;  /* calls that were re-run without device-side waits after one of them gave up -- a barrier of the
                                 fused kernel (its workgroups were not resident together) or a hand-over between the two
                                 streams (they were not running beside each other); the engine then runs one launch per
                                 MINRES step (ICAR) / on one stream (reduced-rank model) until a later call finds the device
                                 as creation found it (repromotions) */
    double profile_minres_iterations; /* mean MINRES iterations per solve over the k_iter launches of the last occ_profile */
    int64_t iter_kernel_launches;     /* k_iter launches of the last occ_run ... */
    double iter_kernel_mean_us;       /* ... and their mean duration, first workgroup in to last chain out, by the
                                         device's constant-rate wall clock read inside the kernel */
    /* ABI 5: streams are pooled per (process, device, CU partition) -- a CU-masked stream holds one of the device's 24
     * hardware queues for itself, and a process that oversubscribes them gets its queues time-sliced */
    int32_t repromotions;         /* returns to the fused kernel / device-side hand-overs after a run-time fallback */
    int32_t stream_probes;        /* "do my two streams run beside each other?" asked so far (creation + whenever the
                                     process's set of streams changed before a call) */
    int32_t handover_mode;        /* 2: device-side counters, 1: events (or one stream), between the two streams */
    int32_t stream_pairs_masked;  /* live CU-masked stream pairs of this process on the handle's device ... */
    int32_t stream_pairs_plain;   /* ... and unmasked ones (each pair is shared by all engines that want its partition) */
    int32_t demoted;              /* 1: running without device-side waits after a fallback (see fused_fallbacks) */
    double profile_iter_dispatch_us; /* mean DISPATCH duration of k_iter over the launches of the last occ_profile: start / stop
                                        events of hipExtLaunchKernel, i.e. the begin / end timestamps of the dispatch's
                                        completion signal -- what rocprofv3 --kernel-trace reports for a kernel (launch ramp
                                        and end-of-kernel release included, which the in-kernel clock cannot see) */
    /* ABI 6 */
    int32_t stream_pairs_idle;    /* pairs no engine holds, kept for the next taker of their CU partition (not live) */
    int32_t stream_pairs_evicted; /* idle CU-masked pairs this process destroyed to make room under the cap of four, all devices */
} occ_stats;
int occ_get_stats(occ_sampler *s, occ_stats *out);

/* Per-kernel launch time in the mode occ_run uses: for each kernel kind, `reps` back-to-back launches
 * of that ONE kernel are captured into a hipGraph and bracketed by two HIP events on the engine's
 * stream; total_us[kind] / counts[kind] = kernel duration + one dependent-launch boundary.  k_minres is
 * timed inside a replayed graph of a real solve prefix (k_eta_init + launches 1..8), see occ_gibbs.hip.
 * kinds: 0 omega_b, 1 noise, 2 eta_init, 3 minres, 4 beta_partial, 5 omega_a, 6 alpha_draw,
 * 7 z_ob (beta draw + z update + next iteration's omega_b), 8 iter (the fused iteration kernel k_iter, timed IN SITU
 * first: `reps` real iterations continue the chains, nothing recorded, two HIP events around every k_iter launch
 * on the main stream while the side stream runs omega_a / alpha / noise as in occ_run; counts[8] = launches,
 * total_us[8] = sum of their durations; zero when the engine does not use the fused iteration).
 * The chains are left mid-solve in an unspecified state: call occ_set_start before sampling again. */
#define OCC_N_KERNEL_KINDS 9
int occ_profile(occ_sampler *s, int32_t reps, int64_t counts[OCC_N_KERNEL_KINDS],
                double total_us[OCC_N_KERNEL_KINDS]);

/* ---- chains sharded over GPUs (reference gibbs/parallel.py:4-42 runs one process per chain) -----------
 * Chains are independent: the only communication is ONE broadcast of the fixed problem arrays at set-up, device to
 * device over RCCL (xGMI), called directly -- librccl is opened with dlopen when one of these entry points is first
 * used -- and never a host round trip on the receiving side.  There is no per-iteration collective.
 *
 * In-process (what LogitICARGibbs(..., devices=[...]).sample(chains=N) does; chain c lives on devices[c % G]):
 *   occ_create_group lays the problem out once, uploads it to devices[0], creates one sampler per device and broadcasts
 *   the fixed arrays from devices[0] (ncclCommInitAll + grouped ncclBroadcast; hipMemcpyPeer if librccl is unusable or
 *   OCC_GROUP_TRANSPORT=peer).  keys: the chains' keys, device after device.  Each handle is then driven by its own
 *   host thread (the C ABI holds no global state; ctypes releases the GIL).
 * One process per GPU (bench.py --gpus N under torch.distributed.run, or any launcher that sets RANK / WORLD_SIZE):
 *   occ_comm_unique_id on rank 0 -> the 128 bytes reach the other ranks by any side channel -> occ_comm_create on every
 *   rank (ncclCommInitRank) -> occ_create_distributed: the root lays the problem out and uploads it, the other ranks
 *   (problem = NULL) size their arrays from a small header and receive them by ncclBroadcast.  occ_comm_barrier /
 *   occ_comm_allreduce_max / occ_comm_broadcast_host are the host-side collectives a benchmark needs (staged through
 *   a device buffer).  occ_group_transport says how a handle got its arrays. */
typedef struct occ_comm occ_comm;
int occ_create_group(const occ_problem *problem, int32_t n_devices, const int32_t *devices, const int32_t *chains_per_device,
                     const uint64_t *keys, occ_sampler **out /* n_devices handles */);
int occ_comm_unique_id(uint8_t id[128]);
int occ_comm_create(int32_t world, int32_t rank, const uint8_t id[128], int32_t device, occ_comm **out);
int occ_comm_destroy(occ_comm *comm);
int occ_comm_barrier(occ_comm *comm);
int occ_comm_allreduce_max(occ_comm *comm, double *inout, int32_t n);
int occ_comm_broadcast_host(occ_comm *comm, void *buf, int64_t bytes, int32_t root);
const char *occ_comm_last_error(const occ_comm *comm); /* NULL: error of the last failed occ_comm_create / _unique_id */
int occ_create_distributed(const occ_problem *problem /* root only */, occ_comm *comm, int32_t root, int32_t n_chains,
                           const uint64_t *keys, occ_sampler **out);
const char *occ_group_transport(const occ_sampler *s);
int occ_synchronize(occ_sampler *s); /* the handle's streams have drained (deadline poll), then hipDeviceSynchronize on its device */

/* ---- per-conditional entry points with INJECTED variates ---------------------------------------------------
 * One conditional update of ONE chain of the ICAR model, run on the device by the kernels of the launch-per-step path
 * (grid over that chain only), with the random variates the reference would have drawn supplied by the caller instead
 * of the chain's Philox streams.  Inputs that are not arguments are the chain's current state (occ_set_start /
 * occ_set_state: alpha, beta, tau, eta, z, xz).  They exist so that a test can feed the inputs and the recorded
 * variates of the REFERENCE's own run (tests/golden/\*.npz) and compare with the reference's recorded outputs in one
 * hop (tests/test_gpu_golden.py); the Polya-Gamma variates (omega_b, omega_a) are inputs, since their draw counts are
 * data dependent.  The chain's iteration number does not advance; results are also left in the chain's state, so the
 * calls chain like the reference's _update_* methods.  Call occ_set_start before sampling with the handle again.
 *
 * occ_cond_tau    logit.py:206-209   tau = gamma_variate / (eta'Q eta / 2 + tau_rate), gamma_variate ~ standard
 *                                    gamma(tau_shape) (numpy: rng.gamma(shape, 1 / rate) = standard_gamma(shape) / rate)
 * occ_cond_eta    logit.py:211-217, 73-99   omega_b[n]; eps_site[n] = the first n standard normals; prior_term[n] = the
 *                                    N(0, Q) vector E eps_2 of logit.py:77 BEFORE the sqrt(tau) factor (the engine draws
 *                                    it in edge form, a test passes the reference's).  Uses the chain's beta, z, tau
 *                                    and warm start xz.  Outputs (each may be NULL): the right-hand side y[n],
 *                                    [x z][2n], eta[n], the MINRES iteration count.
 * occ_cond_beta   logit.py:226-232, distributions.pyx:42-110   omega_b[n], eps[p] standard normals; the chain's eta, z
 * occ_cond_alpha  logit.py:180-190, 219-224   omega_a[R] in flat visit-row order (rows of sites that do not exist --
 *                                    no detection and z = 0 -- are ignored), eps[q]; the chain's z
 * occ_cond_z      logit.py:234-252   u[n]: the uniform of site i (sites with a detection ignore theirs); the chain's
 *                                    alpha, beta, eta.  z_out[n] in {0, 1}. */
int occ_cond_tau(occ_sampler *s, int32_t chain, double gamma_variate, double *tau_out);
int occ_cond_eta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps_site, const double *prior_term,
                 double *rhs_out, double *xz_out, double *eta_out, int32_t *itn_out);
int occ_cond_beta(occ_sampler *s, int32_t chain, const double *omega_b, const double *eps, double *beta_out);
int occ_cond_alpha(occ_sampler *s, int32_t chain, const double *omega_a, const double *eps, double *alpha_out);
int occ_cond_z(occ_sampler *s, int32_t chain, const double *u, double *z_out);

/* Variates of the engine's own generators, drawn ON THE DEVICE by the device functions the kernels use, for the
 * known-answer and distributional tests of the samplers that stand where the reference calls the third-party
 * polyagamma package (logit.py:191-193, 202-204) and numpy's Generator.gamma (logit.py:209): out[i] comes from the
 * sub-stream (key, index i, iteration, stream) exactly as a kernel of the iteration would draw it.
 * kind 0: PG(1, param[i]);  1: standard gamma of shape param[i];  2: standard normal;  3: uniform on (0, 1)
 * (param is ignored for kinds 2 and 3).  Host pointers (or device pointers of `device`); n < 2^31.
 * kind 4 is a device self-test, not a variate: n (a multiple of 64) values in param, out[i] = the wave sum of one of four
 * quantities derived from them, NaN where the three forms of the engine's wave sum (plain, four at once, transposed)
 * disagree in a bit (tests/test_gpu_rng.py).
 * Errors are reported through occ_last_error(NULL). */
int occ_draw(int32_t device, int32_t kind, uint64_t key, uint32_t iteration, uint32_t stream, int64_t n, const double *param,
             double *out);

const char *occ_last_error(const occ_sampler *s); /* NULL handle: error of the last failed occ_create */
int32_t occ_abi_version(void);
int32_t occ_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
