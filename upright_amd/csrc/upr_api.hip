// upr_api.hip -- C-ABI of libupright_mi.so (include/upright_mi.h): host orchestration of the HIP kernels.
// No CPU compute path: every entry point that computes launches kernels on the current HIP device.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "upr_common.h"
#include "upr_kin.h"
#include "upr_linearize.h"
#include "upr_linearize2.h"
#include "upr_linesearch.h"
#include "upr_qp.h"
#include "upr_qp2.h"
#include "upr_qp3.h"
#include "upr_qp3_list.h"
#include "upr_qp3_launch.h"

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}

#define UPR_HIP(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_) + " (upr_api.hip:" + std::to_string(__LINE__) + ")"); \
    } while (0)

// device scratch of one call: freed on every exit path (the early returns of UPR_HIP included)
template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        UPR_HIP(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)));
        UPR_HIP(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
        return 0;
    }
    operator T*() const { return p; }
};

template <class T>
int dev_alloc(T** p, size_t n) {
    UPR_HIP(hipMalloc((void**)p, n * sizeof(T)));
    UPR_HIP(hipMemset(*p, 0, n * sizeof(T)));
    return 0;
}

// ---- small kernels -----------------------------------------------------------------------------------
__global__ void core_object_dynamics_kernel(const upr_problem* P, const double* body_params, int n, const double* forces,
                                            const double* C, const double* w, const double* al, const double* a, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nb = P->nb, nfc = P->nf * P->nc;
    upr_ee<double> E;
    for (int r = 0; r < 9; ++r) E.C[r] = C[(size_t)i * 9 + r];
    for (int r = 0; r < 3; ++r) { E.w[r] = w[(size_t)i * 3 + r]; E.al[r] = al[(size_t)i * 3 + r]; E.a[r] = a[(size_t)i * 3 + r]; E.p[r] = 0; E.v[r] = 0; }
    double Fw[6 * UPR_MAX_BODIES];
    upr_object_wrenches(P, body_params, forces + (size_t)i * nfc, Fw);
    for (int b = 0; b < nb; ++b)
        upr_body_residual<double>(E, body_params + 10 * b, P->gravity, Fw + 6 * b, Fw + 6 * b + 3, out + (size_t)i * 6 * nb + 6 * b);
}

__global__ void core_friction_rows_kernel(const upr_problem* P, int n, const double* forces, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * P->nc) return;
    int s = i / P->nc, c = i % P->nc;
    upr_friction_rows_contact(P, c, forces + ((size_t)s * P->nc + c) * 3, out + ((size_t)s * P->nc + c) * 5);
}

// initial guess of every instance: previous solution re-sampled on the new grid (warm start), or the
// stationary guess x_k = x0, u = 0 (DefaultInitializer); the first node is always the observation.
__global__ void prepare_kernel(const upr_problem* P, upr_dims d, int B, int has_prev /* a previous solution exists (every instance or none) */, const double* tprev,
                               const double* xs_prev, const double* us_prev, const double* t0, const double* x0,
                               double* xs, double* us, double* stats, int* done) {
    int b = blockIdx.x;
    const int N = d.N, nx = d.nx, nu = d.nu;
    const double* xp = xs_prev + (size_t)b * (N + 1) * nx;
    const double* up = us_prev + (size_t)b * N * nu;
    double* xo = xs + (size_t)b * (N + 1) * nx;
    double* uo = us + (size_t)b * N * nu;
    const bool warm = has_prev != 0;
    for (int e = threadIdx.x; e < (N + 1) * nx; e += blockDim.x) {
        int k = e / nx, i = e % nx;
        double v;
        if (k == 0 || !warm) v = x0[(size_t)b * nx + i];
        else upr_interp(d, P->dt, tprev[b], xp, up, t0[b] + k * P->dt, i, 0, &v, nullptr);
        xo[e] = v;
    }
    for (int e = threadIdx.x; e < N * nu; e += blockDim.x) {
        int k = e / nu, i = e % nu;
        double v = 0.0;
        if (warm) upr_interp(d, P->dt, tprev[b], xp, up, t0[b] + k * P->dt, 0, i, nullptr, &v);
        uo[e] = v;
    }
    for (int e = threadIdx.x; e < UPR_NSTATS; e += blockDim.x) stats[(size_t)b * UPR_NSTATS + e] = 0.0;
    if (threadIdx.x == 0) done[b] = 0;
}

__global__ void evaluate_kernel(const upr_problem* P, upr_dims d, int B, const double* tsol, const double* xs,
                                const double* us, const double* t, int t_stride, double* x_out, double* u_out) {
    int b = blockIdx.x;
    const int N = d.N, nx = d.nx, nu = d.nu;
    const double tau = t[(size_t)b * t_stride];
    for (int i = threadIdx.x; i < nx; i += blockDim.x)
        upr_interp(d, P->dt, tsol[b], xs + (size_t)b * (N + 1) * nx, us + (size_t)b * N * nu, tau, i, 0, x_out + (size_t)b * nx + i, nullptr);
    for (int i = threadIdx.x; i < nu; i += blockDim.x)
        upr_interp(d, P->dt, tsol[b], xs + (size_t)b * (N + 1) * nx, us + (size_t)b * N * nu, tau, 0, i, nullptr, u_out + (size_t)b * nu + i);
}

// Where the QP kernel that ran left its per-knot factors (doubles, relative to the instance workspace)
struct upr_fb_src {
    int kind;                       // 3: K stored; 1, 2: V = Lj^-1 Hux and Lji = Lj^-1 stored
    long k_base; int k_stride;      // kind 3: K_k ; kinds 1, 2: V_k
    long lji_base; int lji_stride;  // kinds 1, 2
    long lfi_base; int lfi_stride;  // inverse Cholesky factor of the contact-force blocks of Hff
    long lsi_base; int lsi_stride;  // inverse Cholesky factor of S = Df Hff^-1 Df'
    int lsi_sb;                     // its block size: ne (one dense factor per knot) or 6 (star arrangements of the production kernel: one 6 x 6 block per body)
};

// Linear feedback gains of the last QP, ocs2 sign convention (u = bias + K x): fb[B][N][nu][nx].
// Jerk rows: -Hjj^-1 Hux from the Riccati recursion; contact-force rows: -Hff^-1 Df' S^-1 C (the forces
// follow the state through the object-dynamics equality).
// NEM / NFM: compile-time bounds of ne / nfc (the per-thread scratch vectors stay in registers for the small shapes)
template <int NEM, int NFM>
__global__ void feedback_kernel(const upr_problem* P, upr_dims d, upr_fb_src src, const double* ws, const double* lin,
                                const double* Df, const double* stats, double* fb) {
    const int b = blockIdx.x;
    const int N = d.N, nq = d.nq, nx = d.nx, nu = d.nu, ne = d.ne, nfc = d.nfc, nf = d.nf;
    const double* w = ws + (size_t)b * d.ws_stride;
    double* out = fb + (size_t)b * N * nu * nx;
    if (stats[(size_t)b * UPR_NSTATS + 2] == 2.0) {
        // the last QP's factorisation broke down (status 2, no step taken): its factors are not a policy; the instance
        // keeps its plan as a feed-forward input until the next successful solve
        for (int e = threadIdx.x; e < N * nu * nx; e += blockDim.x) out[e] = 0.0;
        return;
    }
    for (int e = threadIdx.x; e < N * nq * nx; e += blockDim.x) {
        const int k = e / (nq * nx), j = (e % (nq * nx)) / nx, c = e % nx;
        double v;
        if (src.kind == 3) v = w[src.k_base + (long)k * src.k_stride + j * nx + c];
        else {
            const double* V = w + src.k_base + (long)k * src.k_stride;
            const double* Lji = w + src.lji_base + (long)k * src.lji_stride;
            v = 0.0;
            for (int m = j; m < nq; ++m) v += Lji[m * nq + j] * V[m * nx + c];
        }
        out[((size_t)k * nu + j) * nx + c] = -v;
    }
    const double* Dfb = Df + (size_t)b * ne * nfc;
    for (int e = threadIdx.x; e < N * nx; e += blockDim.x) {
        const int k = e / nx, c = e % nx;
        const double* Ck = lin + ((size_t)b * (N + 1) + k) * d.lin_stride + d.lin_gx;
        const double* Lsi = w + src.lsi_base + (long)k * src.lsi_stride;
        const double* Lfi = w + src.lfi_base + (long)k * src.lfi_stride;
        double t1[NEM], t2[NEM], t3[NFM];
        // S^-1 C = Lsi' (Lsi C), block by block (a dense factor is one block of ne rows)
        const int sb = src.lsi_sb;
        for (int r = 0; r < ne; ++r) { const int b0 = (r / sb) * sb; const double* Lb = Lsi + (b0 / sb) * sb * sb; double v = 0.0; for (int m = b0; m <= r; ++m) v += Lb[(r - b0) * sb + (m - b0)] * Ck[m * nx + c]; t1[r] = v; }
        for (int r = 0; r < ne; ++r) { const int b0 = (r / sb) * sb; const double* Lb = Lsi + (b0 / sb) * sb * sb; double v = 0.0; for (int m = r; m < b0 + sb; ++m) v += Lb[(m - b0) * sb + (r - b0)] * t1[m]; t2[r] = v; }
        for (int i = 0; i < nfc; ++i) { double v = 0.0; for (int r = 0; r < ne; ++r) v += Dfb[r * nfc + i] * t2[r]; t3[i] = v; }
        for (int ci = 0; ci < d.nc; ++ci) {
            if (nf == 3) {
                const double* Bk = Lfi + 9 * ci; const double* x3 = t3 + 3 * ci;
                double y[3], z[3];
                for (int a = 0; a < 3; ++a) { double v = 0.0; for (int b2 = 0; b2 <= a; ++b2) v += Bk[3 * a + b2] * x3[b2]; y[a] = v; }
                for (int a = 0; a < 3; ++a) { double v = 0.0; for (int b2 = a; b2 < 3; ++b2) v += Bk[3 * b2 + a] * y[b2]; z[a] = v; }
                for (int a = 0; a < 3; ++a) out[((size_t)k * nu + nq + 3 * ci + a) * nx + c] = -z[a];
            } else {
                const double lf = Lfi[ci];
                out[((size_t)k * nu + nq + ci) * nx + c] = -lf * lf * t3[ci];
            }
        }
    }
}

// u(t, x) = (1 - a) [u_j + K_j (x - x_j)] + a [u_j+1 + K_j+1 (x - x_j+1)]   (ocs2::LinearController with
// bias_j = u_j - K_j x_j, both arrays interpolated linearly); the last interval keeps u_{N-1}, K_{N-1}.
__global__ void evaluate_policy_kernel(const upr_problem* P, upr_dims d, int B, const double* tsol, const double* xs,
                                       const double* us, const double* fb, const double* t, const double* x_obs,
                                       double* x_out, double* u_out) {
    const int b = blockIdx.x;
    const int N = d.N, nx = d.nx, nu = d.nu;
    const double* X = xs + (size_t)b * (N + 1) * nx; const double* U = us + (size_t)b * N * nu;
    const double* K = fb + (size_t)b * N * nu * nx; const double* xo = x_obs + (size_t)b * nx;
    double s = (t[b] - tsol[b]) / P->dt;
    if (s < 0.0) s = 0.0;
    for (int i = threadIdx.x; i < nx; i += blockDim.x)
        upr_interp(d, P->dt, tsol[b], X, U, t[b], i, 0, x_out + (size_t)b * nx + i, nullptr);
    for (int i = threadIdx.x; i < nu; i += blockDim.x) {
        double v = 0.0;
        if (s <= N) {
            int j = (int)s; if (j > N - 1) j = N - 1;
            const double a = (s >= N - 1) ? 0.0 : s - j;
            const int j1 = (j + 1 < N) ? j + 1 : N - 1;
            double v0 = U[j * nu + i], v1 = U[j1 * nu + i];
            for (int c = 0; c < nx; ++c) {
                v0 += K[((size_t)j * nu + i) * nx + c] * (xo[c] - X[j * nx + c]);
                if (a > 0.0) v1 += K[((size_t)j1 * nu + i) * nx + c] * (xo[c] - X[j1 * nx + c]);
            }
            v = (1.0 - a) * v0 + a * v1;
        }
        u_out[(size_t)b * nu + i] = v;
    }
}

}  // namespace

// ========================================================================================================
// every entry point that takes a handle: the handle's own device is made current for the calling thread (the HIP "current
// device" is per thread; a stream may only be used with its device current), so that one process can hold engines on several
// GPUs and a launcher's rank r runs on the GPU its engine was created on whatever another library selected in between
#define UPR_ENTER(h) do { if (!(h)) return fail("null batch"); if ((h)->device >= 0) UPR_HIP(hipSetDevice((h)->device)); } while (0)

// an instantiation of the production QP kernel made at run time (hiprtc; "run-time instantiation" further down)
struct upr_jit_kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    size_t lds = 0, ws = 0;
    int nt = 256;       // lanes per workgroup (UPR_JIT_NT: 128 | 256 | 512, experiments)
    std::string cfg;    // the template arguments, as rocprofv3 prints them
    int fbo[7] = {0, 0, 0, 0, 0, 0, 0};   // where this instantiation leaves K, Lf^-1, Ls^-1 (fb_source): far + Ks, nq nx, far + lfi, NLF, far + lsi, NLS, SB
};

struct upr_batch {
    upr_problem P;
    upr_dims d;
    int B = 0;
    int device = -1;            // HIP device the handle's buffers and stream live on (the current device at upr_batch_create)
    hipStream_t stream = nullptr;
    upr_problem* dP = nullptr;
    double *body_params = nullptr, *way_p = nullptr, *way_q = nullptr, *t0 = nullptr, *x0 = nullptr;
    double *xs = nullptr, *us = nullptr, *xs_prev = nullptr, *us_prev = nullptr, *tprev = nullptr;
    double *lin = nullptr, *Df = nullptr, *ws = nullptr, *stats = nullptr;
    int *done = nullptr;
    bool has_prev = false;   // a solution of a previous advance exists (warm start, policy evaluation)
    unsigned char* iter_key = nullptr;   // [B] IPM iteration count of the last QP, one byte per instance (what `order` is ranked by)
    int* order = nullptr;       // dispatch order of the QP launch: instances by the iteration count of their last QP, longest first
    bool order_on = true, order_valid = false;
    double* prof = nullptr;
    double *ev_t = nullptr, *ev_xo = nullptr, *ev_x = nullptr, *ev_u = nullptr;   // scratch of the evaluate / evaluate_policy calls
    double* kkt = nullptr;   // multiplier export of the register-resident QP kernel (upr_batch_qp_kkt), allocated on first use
    double *fb = nullptr, *xs_lin = nullptr;   // feedback gains of the last solve
    // dynamic obstacle (n_dyn == 1): observed state per instance (device + host copies, and the one the stored
    // solution belongs to), activation flag of the projectile rows
    double *dyn0 = nullptr, *pflag = nullptr;
    std::vector<double> hdyn0, hdyn_prev, htprev;
    double* pin = nullptr;   // pinned host staging of upr_batch_tick
    // upr_batch_tick as a HIP graph: the eleven stream operations of a control period (three copies in, prepare, linearise, QP,
    // line search, policy, three copies out) captured once the handle is in its steady state and replayed while that state --
    // tick_sig: everything on the host that decides WHICH operations a tick enqueues -- stays the same
    hipGraphExec_t tick_exec = nullptr;
    unsigned long long tick_sig = 0;
    int tick_steady = 0;        // ticks in a row with the same signature
    int tick_graph_on = -1;     // UPR_TICK_GRAPH (default 1), read once
    long long tick_replays = 0;
    int nxf = 0;   // interface state dimension 3 nq + 9 n_dyn
    bool guess_set = false;
    int sqp_iters_next = 0;   // > 0: SQP iterations of the next advance only (init_sqp_iteration of the first solve)
    double last_ms = 0.0;
    int qp_nt = 0;
    bool use_qp2 = false;
    int use_qp3 = 0;   // 0: no; 1: headline instantiations; 2: one of UPR_QP3_EXTRA (qp3_variant); 3: instantiated at run time (jit)
    struct upr_jit_kernel* jit = nullptr;
    bool fb_fused = false;   // the selected QP kernel writes the feedback gains itself (upr_qp_args::fb)
    bool use_mfma = true;
    bool lin2 = true;   // shapes without collision rows / orientation cost: upr_linearize2_kernel (UPR_LIN2=0 at create: upr_linearize_kernel)
    int timing = 0;   // 0: no events; 1: around every kernel of an advance; 2: around the QP kernel only; 3: around every FOURTH QP launch
    unsigned timing_qp_count = 0;   // QP launches since upr_batch_enable_timing (mode 3 samples those with count % 4 == 0)
    double k_ms[3] = {0, 0, 0};
    int k_launches[3] = {0, 0, 0};
    std::vector<double> hDf;
    std::vector<double> held_stats; std::vector<unsigned char> held_keys;   // upr_batch_hold_stats
    std::vector<double> kkt_slack;   // slacks of the rows at the exit of the last upr_batch_qp_kkt ([B][N+1][ni]; upr_batch_qp_slacks)
    std::string qp_name;   // the QP kernel instantiation this handle launches
    std::vector<hipEvent_t> ev_pool, ev_free;   // events in use (pairs, in launch order) / harvested ones waiting for reuse
    std::vector<int> ev_slot;
};

namespace {

int check_problem(const upr_problem* P) {
    if (!P) return fail("upr_problem is NULL");
    if (P->nq != 6 && P->nq != 9) return fail("unsupported chain length nq = " + std::to_string(P->nq) + " (kernels are instantiated for 6 and 9 joints)");
    if (P->nb < 1 || P->nb > UPR_MAX_BODIES) return fail("nb out of range");
    if (P->nc < 1 || P->nc > UPR_MAX_CONTACTS) return fail("nc out of range");
    if (P->nf != 1 && P->nf != 3) return fail("nf must be 1 or 3");
    if (P->N < 1 || P->N > 1000) return fail("N out of range");
    if (!(P->dt > 0)) return fail("dt must be positive");
    if (P->n_way < 1 || P->n_way > UPR_MAX_WAYPOINTS) return fail("n_way out of range");
    for (int i = 0; i < 6; ++i) if (!(P->Wee[i] >= 0)) return fail("end-effector weights must be non-negative");
    if (P->n_sph < 0 || P->n_sph > UPR_MAX_SPHERES) return fail("n_sph out of range");
    if (P->n_pairs < 0 || P->n_pairs > UPR_MAX_PAIRS) return fail("n_pairs out of range");
    if (P->n_dyn < 0 || P->n_dyn > UPR_MAX_DYN) return fail("n_dyn must be in 0 .. UPR_MAX_DYN");
    if (P->n_proj < 0 || P->n_proj > 8 || (P->n_proj > 0 && P->n_dyn < 1)) return fail("projectile rows need a dynamic obstacle (n_proj <= 8; they follow the last one)");
    if (P->soft_state_box || P->soft_input_box || P->soft_poly) {
        if (!(P->soft_L2_lower >= 0) || !(P->soft_L2_upper >= 0) || !(P->soft_L1_lower >= 0) || !(P->soft_L1_upper >= 0)) return fail("slack penalties must be non-negative");
        if (!(P->soft_L2_lower + P->soft_L1_lower > 0) || !(P->soft_L2_upper + P->soft_L1_upper > 0)) return fail("softened rows need a positive L1 or L2 penalty");
    }
    if (P->soft_eq && (!(P->soft_L2_lower > 0) || P->soft_L2_lower != P->soft_L2_upper || P->soft_L1_lower != 0 || P->soft_L1_upper != 0))
        return fail("a softened object-dynamics equality needs equal positive L2 penalties and zero L1 penalties (the slack pair is eliminated to a quadratic penalty)");
    for (int i = 0; i < P->n_proj; ++i) if (P->proj_sph[i] < 0 || P->proj_sph[i] >= P->n_sph || !(P->proj_dist[i] > 0)) return fail("projectile row out of range");
    for (int i = 0; i < P->n_sph; ++i) if (P->sph_frame[i] > P->nq || (P->sph_frame[i] <= -2 && -2 - P->sph_frame[i] >= P->n_dyn)) return fail("sph_frame out of range");
    for (int i = 0; i < P->n_pairs; ++i)
        if (P->pair_a[i] < 0 || P->pair_a[i] >= P->n_sph || P->pair_b[i] < -1 || P->pair_b[i] >= P->n_sph || P->pair_a[i] == P->pair_b[i]) return fail("collision pair out of range");
    for (int i = 0; i < P->nc; ++i) {
        if (P->contact_body2[i] < 0 || P->contact_body2[i] >= P->nb) return fail("contact_body2 must index a balanced body");
        if (P->contact_body1[i] >= P->nb) return fail("contact_body1 out of range");
    }
    return 0;
}

int need_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) return fail("no HIP device available: libupright_mi has no CPU path");
    return 0;
}

#ifndef UPR_LIN_ROW_PASSES_DEFAULT
#define UPR_LIN_ROW_PASSES_DEFAULT 2
#endif
template <int NQ>
int launch_linearize(upr_batch* h, const upr_lin_args& A) {
    static const int occ = getenv("UPR_LIN_OCC") ? atoi(getenv("UPR_LIN_OCC")) : 2;
    // knots per workgroup: several passes (their value walks side by side) for the plain instantiation without collision rows
    const bool multi = UPR_LIN_ANALYTIC && !A.way_q && h->use_mfma && occ == 2 && A.d.no == 0;
    // with collision rows (snapshot form, round 4: 156 instead of 480 doubles of LDS per knot for 16 spheres): UPR_LIN_ROW_PASSES
    // passes per workgroup (1, 2 or 3)
    static const int row_passes = getenv("UPR_LIN_ROW_PASSES") ? atoi(getenv("UPR_LIN_ROW_PASSES")) : UPR_LIN_ROW_PASSES_DEFAULT;
    const bool multi_rows = UPR_LIN_ANALYTIC && UPR_LIN_OBS_SNAP && !A.way_q && h->use_mfma && occ == 2 && A.d.no > 0 && (row_passes == 2 || row_passes == 3);
    // no collision rows, no orientation cost: the one-round kernel of upr_linearize2.h, as many knots per workgroup as three
    // workgroups per CU hold in LDS and one tangent pass takes in one trip (28 for the headline shape: 768 workgroups)
    if (h->lin2 && upr_lin2_eligible(A) && h->use_mfma && occ == 2) {
        const upr_lin2_lay lay = upr_lin2_layout(A.d, h->P.n_sph);
        const int npre = ((UPR_LIN2_NPRE + 1) & ~1) + (A.d.no > 0 ? upr_lin2_table_doubles(h->P.n_sph) : 0);
        int kpw = (int)((UPR_LIN2_LDS_BUDGET - npre * sizeof(double)) / (lay.per * sizeof(double)));
        if (kpw > 256 / NQ) kpw = 256 / NQ;
        if (kpw > 64) kpw = 64;
        if (kpw >= 1) {
            const size_t lds2 = (size_t)(npre + kpw * lay.per) * sizeof(double);
            hipLaunchKernelGGL(upr_linearize2_kernel<NQ>, dim3((A.npoints + kpw - 1) / kpw), dim3(256), lds2, h->stream, A, kpw, h->P.n_sph);
            UPR_HIP(hipGetLastError());
            return 0;
        }
    }
    const int KPW = 8 * (multi ? UPR_LIN_PASSES : (multi_rows ? row_passes : 1));
    const int blocks = (A.npoints + KPW - 1) / KPW;
    const size_t lds = (size_t)KPW * upr_lin_lds_doubles(A.d, h->P.n_sph) * sizeof(double) + sizeof(upr_problem) + 16;   // (+ the kernel's copy of the problem record)
    if (lds > 160 * 1024) return fail("collision model too large for the linearisation kernel's LDS");
    // (rounds 1 - 2, one forward-mode walk per tangent lane: 2 -> 0.130 ms, 3 -> 0.142 ms, 4 -> 0.32 ms with spills; round 3, one value
    // walk per knot + closed-form tangents: 2 -> 0.091 ms, 3 -> 0.085 ms;
    // with the walks of three passes side by side: 2 -> 0.073 ms, 3 -> 0.095 ms)
    auto launch = [&](void (*kern)(upr_lin_args)) -> int {
        if (lds > 64 * 1024) UPR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, h->stream, A);
        return 0;
    };
    int rc;
    if (A.way_q) rc = h->use_mfma ? launch(upr_linearize_kernel<NQ, true, 2, true>) : launch(upr_linearize_kernel<NQ, false, 2, true>);   // end-effector cost with orientation weights
    else if (!h->use_mfma) rc = launch(upr_linearize_kernel<NQ, false>);
    else if (occ == 3) rc = launch(upr_linearize_kernel<NQ, true, 3>);
    else if (occ == 4) rc = launch(upr_linearize_kernel<NQ, true, 4>);
    else if (multi) rc = launch(upr_linearize_kernel<NQ, true, 2, false, UPR_LIN_PASSES>);
    else if (multi_rows && row_passes == 3) rc = launch(upr_linearize_kernel<NQ, true, 2, false, 3>);
    else if (multi_rows) rc = launch(upr_linearize_kernel<NQ, true, 2, false, 2>);
    else rc = launch(upr_linearize_kernel<NQ, true>);
    if (rc) return rc;
    UPR_HIP(hipGetLastError());
    return 0;
}

// generic (runtime-dimension) QP kernel.  Its LDS footprint (one knot's matrices) allows one or two workgroups per CU for
// the multi-body shapes, so the workgroup size IS the occupancy: 64 lanes left 3 of 4 SIMDs of a CU idle.  NT is chosen by
// the size of the per-knot phases (ne x nx, nx x nx, ne x ne items); UPR_QP_GENERIC_NT overrides it.
template <int NT>
int launch_qp_generic_nt(upr_batch* h, const upr_qp_args& A) {
    const upr_qp_lds lay = upr_qp_lds_layout(A.d, NT);
    const size_t lds = (size_t)lay.total * sizeof(double);
    if (lds > 160 * 1024) return fail("QP working set exceeds 160 KiB of LDS");
    if (lds > 64 * 1024) UPR_HIP(hipFuncSetAttribute((const void*)upr_qp_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(upr_qp_kernel<NT>, dim3(h->B), dim3(NT), lds, h->stream, A);
    UPR_HIP(hipGetLastError());
    return 0;
}
int generic_nt(const upr_batch* h) {
    if (const char* e = getenv("UPR_QP_GENERIC_NT")) return atoi(e);
    const upr_dims& d = h->d;
    const int big = d.ne * d.nx > d.nx * d.nx ? d.ne * d.nx : d.nx * d.nx;
    return big >= 1024 ? 512 : (big >= 512 ? 256 : 64);
}
int launch_qp_generic(upr_batch* h, const upr_qp_args& A) {
    switch (generic_nt(h)) {
        case 64: return launch_qp_generic_nt<64>(h, A);
        case 128: return launch_qp_generic_nt<128>(h, A);
        case 256: return launch_qp_generic_nt<256>(h, A);
        case 512: return launch_qp_generic_nt<512>(h, A);
        case 1024: return launch_qp_generic_nt<1024>(h, A);
        default: return fail("UPR_QP_GENERIC_NT must be 64, 128, 256, 512 or 1024");
    }
}

template <class D, int NT>
int launch_qp2_nt(upr_batch* h, const upr_qp_args& A) {
    const size_t lds = upr_qp2_lds_doubles<D>(A.d.N, NT) * sizeof(double);
    if (lds > 160 * 1024) return fail("QP working set exceeds 160 KiB of LDS");
    if (lds > 64 * 1024) UPR_HIP(hipFuncSetAttribute((const void*)upr_qp2_kernel<D, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((upr_qp2_kernel<D, NT>), dim3(h->B), dim3(NT), lds, h->stream, A);
    UPR_HIP(hipGetLastError());
    return 0;
}
template <class D>
int launch_qp2(upr_batch* h, const upr_qp_args& A) {
    switch (h->qp_nt) {
        case 64: return launch_qp2_nt<D, 64>(h, A);
        case 512:
        case 128: return launch_qp2_nt<D, 128>(h, A);
        case 256: return launch_qp2_nt<D, 256>(h, A);
        default: return fail("UPR_QP_NT must be 64, 128 or 256");
    }
}

// shapes the production kernel is instantiated for: (nq, nb, nc, nf)
// -DUPR_HEADLINE_ONLY: experiment builds of the headline kernel (a tenth of the compile time); never the production library
#ifdef UPR_HEADLINE_ONLY
#define UPR_QP2_SHAPES(X) X(9, 1, 4, 3)
#else
#define UPR_QP2_SHAPES(X) X(9, 1, 4, 3) X(9, 1, 4, 1) X(6, 1, 4, 1) X(6, 1, 4, 3) X(9, 2, 8, 3)
#endif

bool qp2_has_shape(const upr_problem& P) {
#define X(a, b, c, e) if (P.nq == a && P.nb == b && P.nc == c && P.nf == e) return true;
    UPR_QP2_SHAPES(X)
#undef X
    return false;
}
size_t qp2_ws_doubles(const upr_problem& P, const upr_dims& d) {
#define X(a, b, c, e) if (P.nq == a && P.nb == b && P.nc == c && P.nf == e) return upr_qp2_ws_doubles<upr_qp2_dims<a, b, c, e>>(d.N, d.neN);
    UPR_QP2_SHAPES(X)
#undef X
    return 0;
}

// third-structure ("production") kernel.  The headline shape (nq 9, nb 1, nc 4, nf 3, N 20) in three workgroup sizes, with
// and without state-polytopic rows; further (shape, SOFT) instantiations at 256 lanes for the configurations the
// reference ships with HPIPM slacks: thing_demo (one body, frictionless) and the upright_robust 8-corner arrangement.
// (the kernels themselves are instantiated in upr_qp3_inst.hip, one translation unit per part of upr_qp3_list.h, compiled side
// by side; this file only declares their launchers -- -DUPR_MONOLITHIC instantiates them here instead: experiment builds)
template <class C>
int launch_qp3_cfg(upr_batch* h, const upr_qp_args& A) {
    const int rc = upr_qp3_launch<C>(h->stream, h->B, A);
    if (rc == -1) return fail("QP working set exceeds 160 KiB of LDS");
    if (rc != 0) return fail(std::string("upr_qp3_kernel launch: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}
bool qp3_is_headline(const upr_problem& P) { return P.nq == 9 && P.nb == 1 && P.nc == 4 && P.nf == 3 && P.N == 20; }
bool soft_boxes(const upr_problem& P) { return P.soft_state_box || P.soft_input_box; }
// does the problem need a SOFT instantiation?  Slacks on its boxes, or slacks.poly_ineq with friction / state-polytopic rows
bool needs_soft(const upr_problem& P, const upr_dims& d) { return soft_boxes(P) || (P.soft_poly && (d.np > 0 || d.no > 0)); }
// does instantiation (a, b, c, e, horizon n, rows, sf, dense) of UPR_QP3_EXTRA take this problem?  (The first match in list order is the one
// every site below uses: a SOFT problem without state-polytopic rows takes the instantiation without them.)
bool qp3_match(const upr_problem& P, const upr_dims& d, int a, int b, int c, int e, int n, bool rows, bool sf, bool dense) {
    bool star = true;                                         // no two bodies share a contact point
    for (int i = 0; i < P.nc; ++i) if (P.contact_body1[i] >= 0) star = false;
    return P.nq == a && P.nb == b && P.nc == c && P.nf == e && P.N == n && (rows || d.no == 0) && (sf || !needs_soft(P, d)) && (dense || star);
}
// can the production kernel take this problem, and in which instantiation?  0: no; 1: headline (hard boxes); 2: one of
// UPR_QP3_EXTRA
int qp3_variant(const upr_problem& P, const upr_dims& d) {
    if (d.no > UPR_QP3_NOMAX) return 0;
    // (a hard equality the contact forces cannot span -- frictionless one-body arrangements: nf nc < 6 nb -- gets the proximal
    // treatment of upr_qp.h inside the kernel; multi-body shapes of that kind need soft_eq)
    if (d.nfc < d.ne && !P.soft_eq && P.nb > 1) return 0;
    if (qp3_is_headline(P) && !needs_soft(P, d)) return 1;
#define X(a, b, c, e, n, rows, sf, dense) if (qp3_match(P, d, a, b, c, e, n, rows, sf, dense)) return 2;
    UPR_QP3_EXTRA(X)
#undef X
    return 0;
}
template <int NT, bool ROWS>
int launch_qp3_rows(upr_batch* h, const upr_qp_args& A) { return launch_qp3_cfg<upr_qp3_cfg<9, 1, 4, 3, 20, NT, ROWS>>(h, A); }
// problems without state-polytopic rows run the instantiation that has none compiled in (upr_qp3.h, upr_qp3_cfg)
template <int NT>
int launch_qp3(upr_batch* h, const upr_qp_args& A) {
    return (h->d.no > 0) ? launch_qp3_rows<NT, true>(h, A) : launch_qp3_rows<NT, false>(h, A);
}
size_t qp3_ws_doubles(const upr_problem& P, const upr_dims& d, int variant) {
    if (variant == 1) return upr_qp3_ws<upr_qp3_cfg<9, 1, 4, 3, 20, 256>>::total;
#define X(a, b, c, e, n, rows, sf, dense) if (qp3_match(P, d, a, b, c, e, n, rows, sf, dense)) return upr_qp3_ws<upr_qp3_cfg<a, b, c, e, n, 256, rows, sf, dense>>::total;
    UPR_QP3_EXTRA(X)
#undef X
    return 0;
}

// ---- run-time instantiation of the production kernel (round 4) ---------------------------------------------------------------
// The hand list of upr_qp3_list.h covers the shapes the reference's configs use.  Any other shape the structure can take (one
// chain of 6 or 9 joints, star or stacked arrangement, nf 1 | 3, with or without rows / slacks, a horizon whose working set fits
// the LDS) is instantiated at upr_batch_create with hiprtc out of the SAME header -- upr_qp3.h with its template arguments as a
// macro -- instead of falling back to the second-structure or generic kernel (20 - 40x slower per QP).  The code object is cached
// in memory per process and on disk ($UPR_JIT_CACHE or ~/.cache/upright_amd) under the shape key and a hash of the kernel
// headers, so a shape costs its compile time once per machine -- measured: about TWO SECONDS (comgr in-process; the same
// instantiation takes hipcc 45 s, most of it the host pass), and the code runs as fast as the ahead-of-time one (headline 2.237
// against 2.224 ms per launch, upright_robust 6.31 against 6.35).  UPR_QP3_JIT = 0: never (the second-structure / generic kernels
// take such shapes, as in rounds 1 - 3); = 2: also for the listed shapes (diagnostic); unset or 1: every capable shape.
std::mutex g_jit_mutex;
std::map<std::string, upr_jit_kernel> g_jit;

extern "C" const char* upr_last_error(void);
std::string jit_csrc_dir() {
    if (const char* e = getenv("UPR_CSRC_DIR")) return e;
    Dl_info info;
    if (dladdr((const void*)&upr_last_error, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t s = p.rfind('/');
        return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/csrc";
    }
    return "upright_amd/csrc";
}
bool jit_read(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss; ss << f.rdbuf(); *out = ss.str();
    return true;
}
// can the production structure take this problem at all?  (what upr_qp3.h asserts or assumes; the LDS is checked after the compile)
bool jit_capable(const upr_problem& P, const upr_dims& d) {
    if (P.nq != 6 && P.nq != 9) return false;
    // (horizons beyond 64 knots: the far-array form of the kernel, upr_qp3_cfg::KFAR -- star arrangements without friction and rows)
    if (d.no > UPR_QP3_NOMAX || P.N < 2 || P.N > 128) return false;
    if (P.N > 64) { for (int i = 0; i < P.nc; ++i) if (P.contact_body1[i] >= 0) return false; if (P.nb < 2 || P.nf != 1 || d.no != 0) return false; }
    if (d.nfc < d.ne && !P.soft_eq && P.nb > 1) return false;
    return true;
}
bool jit_wanted(int B) {
    (void)B;
    if (const char* e = getenv("UPR_QP3_JIT")) return atoi(e) != 0;
    return true;
}
int jit_get(const upr_problem& P, const upr_dims& d, upr_jit_kernel** out) {
    bool star = true;
    for (int i = 0; i < P.nc; ++i) if (P.contact_body1[i] >= 0) star = false;
    const bool rows = d.no > 0, soft = needs_soft(P, d), dense = !star;
    char cfg[128];
    int nt = 256;
    if (const char* e = getenv("UPR_JIT_NT")) { nt = atoi(e); if (nt != 128 && nt != 256 && nt != 512) return fail("UPR_JIT_NT must be 128, 256 or 512"); }
    snprintf(cfg, sizeof(cfg), "%d, %d, %d, %d, %d, %d, %s, %s, %s", P.nq, P.nb, P.nc, P.nf, P.N, nt, rows ? "true" : "false", soft ? "true" : "false", dense ? "true" : "false");
    int dev = 0;
    UPR_HIP(hipGetDevice(&dev));
    const std::string key = std::string(cfg) + " @" + std::to_string(dev);
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit.find(key);
    if (it != g_jit.end()) { *out = &it->second; return 0; }
    const std::string dir = jit_csrc_dir();
    // the translation unit and its options
    const std::string src =
        "#include \"upr_qp3.h\"\n"
        "typedef upr_qp3_cfg<UPR_QP3_JIT_CFG> upr_jit_cfg;\n"
        "extern \"C\" __global__ void __launch_bounds__(upr_jit_cfg::NT, (upr_jit_cfg::NT <= 256 && upr_jit_cfg::NB == 1) ? UPR_QP3_OCC1 : 1) upr_qp3_jit(upr_qp_args A) {\n"
        "    extern __shared__ __attribute__((aligned(16))) double smem[];\n"
        "    upr_ctx ctx; ctx.tid = threadIdx.x; ctx.nt = upr_jit_cfg::NT;\n"
        "    upr_qp3_solve<upr_jit_cfg>(ctx, A, upr_qp_instance(A, blockIdx.x), smem);\n"
        "}\n"
        "extern \"C\" __global__ void upr_qp3_jit_info(int* out) {\n"
        "    typedef upr_qp3_ws<upr_jit_cfg> W; typedef upr_qp3_far<upr_jit_cfg> F;\n"
        "    out[0] = upr_qp3_lds<upr_jit_cfg>::total; out[1] = W::total;\n"
        "    out[2] = W::far + F::Ks; out[3] = upr_jit_cfg::NQ * upr_jit_cfg::NX; out[4] = W::far + F::lfi; out[5] = upr_jit_cfg::NLF;\n"
        "    out[6] = W::far + F::lsi; out[7] = upr_jit_cfg::NLS; out[8] = upr_jit_cfg::SB;\n"
        "}\n";
    const std::string inc = "-I" + dir;
    std::string defc = std::string("-DUPR_QP3_JIT_CFG=") + cfg;
    defc.erase(std::remove(defc.begin(), defc.end(), ' '), defc.end());
    const std::string rinc = std::string("-I") + (getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm") + "/include";
    std::vector<std::string> optv = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed", inc, rinc, defc};
    // UPR_JIT_FLAGS: extra compiler options for experiments (e.g. "-DUPR_QP3_PROF_FLAT=3")
    if (const char* e = getenv("UPR_JIT_FLAGS")) { std::stringstream ss(e); std::string w; while (ss >> w) optv.push_back(w); }
    // version of the code object: FNV-1a over every header the kernel is made of (the C-ABI header with the upr_problem layout
    // among them -- ADVICE r04: a cached kernel compiled for an older layout would read the new struct wrongly), the wrapper
    // above, the options (the include directories excepted: they name a place, not a content) and the compiler's version
    unsigned long long hsh = 1469598103934665603ull;
    auto mix = [&](const std::string& t) { for (unsigned char c : t) { hsh ^= c; hsh *= 1099511628211ull; } hsh ^= 0xff; hsh *= 1099511628211ull; };
    for (const char* f : {"../../include/upright_mi.h", "upr_common.h", "upr_kin.h", "upr_qp.h", "upr_qp2.h", "upr_qp3.h"}) {
        std::string txt;
        if (!jit_read(dir + "/" + f, &txt)) return fail("run-time instantiation: cannot read " + dir + "/" + f + " (set UPR_CSRC_DIR)");
        mix(txt);
    }
    mix(src);
    for (const std::string& o : optv) if (o != inc && o != rinc) mix(o);
    { int vmaj = 0, vmin = 0; (void)hiprtcVersion(&vmaj, &vmin); mix(std::to_string(vmaj) + "." + std::to_string(vmin)); }
    // disk cache: $UPR_JIT_CACHE, else ~/.cache/upright_amd; none when neither is known (no predictable world-writable default)
    std::string cache;
    if (const char* e = getenv("UPR_JIT_CACHE")) cache = e;
    else if (const char* hm = getenv("HOME")) { if (*hm) cache = std::string(hm) + "/.cache/upright_amd"; }
    std::string path;
    if (!cache.empty()) {
        (void)mkdir((cache.substr(0, cache.rfind('/'))).c_str(), 0755);
        (void)mkdir(cache.c_str(), 0755);
        std::string fname = cfg;
        for (char& c : fname) if (c == ',' || c == ' ') c = '_';
        char hx[32]; snprintf(hx, sizeof(hx), "%016llx", hsh);
        path = cache + "/qp3_" + fname + "_" + hx + ".hsaco";
    }
    upr_jit_kernel k;
    k.cfg = cfg; k.nt = nt;
    std::string code;
    bool loaded = false;
    // cache file = [magic "UPRJIT01"][length of the code object][FNV-1a of it][code object]: a truncated or foreign file is recognised
    // HERE -- the loader does not return an error for a cut-off ELF, it crashes -- dropped, and the shape compiled anew
    auto fnv = [](const char* p, size_t n) { unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h; };
    if (!path.empty() && jit_read(path, &code)) {
        bool good = code.size() > 24 && memcmp(code.data(), "UPRJIT01", 8) == 0;
        unsigned long long len = 0, sum = 0;
        if (good) { memcpy(&len, code.data() + 8, 8); memcpy(&sum, code.data() + 16, 8); good = len == code.size() - 24 && sum == fnv(code.data() + 24, (size_t)len); }
        if (good) code.erase(0, 24);
        if (good && hipModuleLoadData(&k.mod, code.data()) == hipSuccess) loaded = true;
        else { (void)hipGetLastError(); (void)unlink(path.c_str()); code.clear(); }
    }
    if (!loaded) {
        fprintf(stderr, "libupright_mi: instantiating the production QP kernel for upr_qp3_cfg<%s> (once per machine: %s)\n", cfg, path.empty() ? "no disk cache" : path.c_str());
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "upr_qp3_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return fail("hiprtcCreateProgram failed");
        std::vector<const char*> opts;
        for (const std::string& o : optv) opts.push_back(o.c_str());
        const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
        if (rc != HIPRTC_SUCCESS) {
            size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
            std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
            hiprtcDestroyProgram(&prog);
            return fail("run-time instantiation of upr_qp3_cfg<" + std::string(cfg) + "> failed: " + log.substr(0, 2000));
        }
        size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
        code.resize(cs); hiprtcGetCode(prog, &code[0]);
        hiprtcDestroyProgram(&prog);
        if (!path.empty()) {   // written next to its final name and renamed only when every byte arrived (a full disk leaves no stub behind)
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
            bool ok;
            {
                const unsigned long long len = code.size(), sum = fnv(code.data(), code.size());
                std::ofstream f(tmp, std::ios::binary);
                f.write("UPRJIT01", 8); f.write((const char*)&len, 8); f.write((const char*)&sum, 8);
                f.write(code.data(), (std::streamsize)code.size()); f.flush(); ok = f.good();
            }
            if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
        }
        UPR_HIP(hipModuleLoadData(&k.mod, code.data()));
    }
    hipFunction_t info;
    UPR_HIP(hipModuleGetFunction(&k.fn, k.mod, "upr_qp3_jit"));
    UPR_HIP(hipModuleGetFunction(&info, k.mod, "upr_qp3_jit_info"));
    DevBuf<int> di;
    if (di.alloc(9)) return 1;
    int* dp = di;
    void* args[] = {&dp};
    UPR_HIP(hipModuleLaunchKernel(info, 1, 1, 1, 1, 1, 1, 0, nullptr, args, nullptr));
    int hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    UPR_HIP(hipMemcpy(hi, dp, sizeof(hi), hipMemcpyDeviceToHost));
    k.lds = (size_t)hi[0] * sizeof(double); k.ws = (size_t)hi[1];
    for (int i = 0; i < 7; ++i) k.fbo[i] = hi[2 + i];
    if (k.lds > 160 * 1024) { (void)hipModuleUnload(k.mod); return fail("run-time instantiation: the working set of upr_qp3_cfg<" + std::string(cfg) + "> exceeds 160 KiB of LDS"); }
    if (k.lds > 64 * 1024) UPR_HIP(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds));
    auto ins = g_jit.emplace(key, k);
    *out = &ins.first->second;
    return 0;
}

int launch_qp(upr_batch* h, const upr_qp_args& A) {
    if (h->use_qp3 == 3) {
        upr_qp_args Ac = A;
        void* args[] = {&Ac};
        UPR_HIP(hipModuleLaunchKernel(h->jit->fn, (unsigned)h->B, 1, 1, (unsigned)h->jit->nt, 1, 1, (unsigned)h->jit->lds, h->stream, args, nullptr));
        return 0;
    }
    if (h->use_qp3 == 1) {
        switch (h->qp_nt) {
            case 256: return launch_qp3<256>(h, A);
            default: return fail("the production QP kernel ships at 256 lanes (UPR_QP_NT = 128 | 512 were A/B instantiations of rounds 1 - 5; UPR_QP3_JIT=2 UPR_JIT_NT=... instantiates one at run time)");
        }
    }
    if (h->use_qp3 == 2) {
#define X(a, b, c, e, n, rows, sf, dense) if (qp3_match(h->P, h->d, a, b, c, e, n, rows, sf, dense)) return launch_qp3_cfg<upr_qp3_cfg<a, b, c, e, n, 256, rows, sf, dense>>(h, A);
        UPR_QP3_EXTRA(X)
#undef X
    }
    if (h->use_qp2) {
#define X(a, b, c, e) if (h->P.nq == a && h->P.nb == b && h->P.nc == c && h->P.nf == e) return launch_qp2<upr_qp2_dims<a, b, c, e>>(h, A);
        UPR_QP2_SHAPES(X)
#undef X
    }
    return launch_qp_generic(h, A);
}

template <int NQ>
int launch_linesearch(upr_batch* h, const upr_ls_args& A0) {
    upr_ls_args A = A0;
    A.n_way = h->P.n_way;
    static_assert(3 * UPR_MAX_WAYPOINTS <= 128, "a lane per waypoint coordinate");
    size_t lds = (size_t)upr_ls_lds_doubles(h->d) * sizeof(double) + sizeof(upr_problem) + 16;   // (+ the kernel's copy of the problem record)
    // Three staged copies (trajectory, step, trial trajectory) pay for the small shapes only: with one wave per workgroup the LDS
    // footprint IS the occupancy, and for the large input vectors it costs more than the uncoalesced reads it replaces
    // (r04, tools/r4_lin.sh: configs[2] 1.58 ms staged / 1.01 ms not, configs[3] 0.249 / 0.171; the headline shape, 40 KB
    // staged, stays).  UPR_LS_STAGE_FULL = 0 / 1 forces either form (A/B runs).
    static const int stage_env = getenv("UPR_LS_STAGE_FULL") ? atoi(getenv("UPR_LS_STAGE_FULL")) : -1;
    const bool stage_off = stage_env == 0 || (stage_env < 0 && lds > 41 * 1024);
    if (lds > 64 * 1024 || stage_off) {   // (also: long horizons of the large shapes, where three copies do not fit)
        A.stage_full = 0;
        lds = (size_t)upr_ls_lds_doubles(h->d, false) * sizeof(double) + sizeof(upr_problem) + 16;
        if (lds > 160 * 1024) return fail("horizon too long for the line-search kernel's LDS");
    }
    // two waves per instance (wave 0: the chain walks of a trial, wave 1: its flat sums): 241 registers, four workgroups per CU by
    // LDS.  Four waves (more flat lanes, 128 registers a lane for four workgroups per CU: 135 spilled) measured 48 against 33 us.
    const int nt = 128;
    auto launch = [&](void (*kern)(upr_ls_args)) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(h->B), dim3(nt), lds, h->stream, A);
    };
    // small shapes (one body, up to four frictional contacts): per-lane vectors sized for them
    // exactly the headline's contact structure (one body on the tray, four frictional contacts): every bound a constant
    const bool st = A.stage_full != 0;   // (compile-time in the kernel: its staged arrays are LDS pointers, not generic ones)
    if (h->d.nfc == 12 && h->d.nb == 1 && h->P.nf == 3 && h->P.nc == 4 && h->d.no == 0) launch(st ? upr_linesearch_kernel<NQ, 128, 12, 1, true, false, true> : upr_linesearch_kernel<NQ, 128, 12, 1, true, false, false>);
    // ... the same with collision / projectile rows (configs[4], the obstacle experiments: round 5)
    else if (h->d.nfc == 12 && h->d.nb == 1 && h->P.nf == 3 && h->P.nc == 4) launch(st ? upr_linesearch_kernel<NQ, 128, 12, 1, true, true, true> : upr_linesearch_kernel<NQ, 128, 12, 1, true, true, false>);
    else if (h->d.nfc <= 12 && h->d.nb == 1) launch(st ? upr_linesearch_kernel<NQ, 128, 12, 1, false, true, true> : upr_linesearch_kernel<NQ, 128, 12, 1, false, true, false>);
    else launch(st ? upr_linesearch_kernel<NQ, 128, 3 * UPR_MAX_CONTACTS, UPR_MAX_BODIES, false, true, true> : upr_linesearch_kernel<NQ, 128, 3 * UPR_MAX_CONTACTS, UPR_MAX_BODIES, false, true, false>);
    UPR_HIP(hipGetLastError());
    return 0;
}

upr_lin_args traj_lin_args(upr_batch* h) {
    upr_lin_args A;
    A.P = h->dP; A.d = h->d; A.body_params = h->body_params; A.way_p = h->way_p; A.way_q = upr_has_orientation_cost(&h->P) ? h->way_q : nullptr; A.t0 = h->t0;
    A.xs = h->xs; A.us = h->us; A.inst = nullptr; A.lin = h->lin; A.ee_out = nullptr;
    A.npoints = h->B * (h->d.N + 1);
    A.dyn = h->dyn0; A.pflag = h->pflag; A.Df = h->Df;
    return A;
}
upr_qp_args make_qp_args(upr_batch* h) {
    upr_qp_args A;
    A.P = h->dP; A.d = h->d; A.xs = h->xs; A.us = h->us; A.x0 = h->x0; A.lin = h->lin; A.Df = h->Df; A.ws = h->ws; A.stats = h->stats; A.prof = h->prof; A.iter_key = h->iter_key;
    return A;
}

int do_linearize(upr_batch* h, const upr_lin_args& A) {
    return (h->P.nq == 6) ? launch_linearize<6>(h, A) : launch_linearize<9>(h, A);
}

// Per-launch device timing with HIP events recorded on the engine's stream (no host sync inside the
// timed region); upr_batch_kernel_times() reads them back after a stream sync.
struct KernelTimer {
    upr_batch* h; int slot; size_t idx; bool on;
    KernelTimer(upr_batch* h_, int s) : h(h_), slot(s), idx(0), on(h_->timing == 1 || (h_->timing == 2 && s == 1)) {
        if (h->timing == 3 && s == 1) on = (h->timing_qp_count++ % 4u) == 0u;
        if (!on) return;
        hipEvent_t a, b;
        // (reused: upr_batch_enable_timing stocks the free list, so that a timed loop creates none)
        if (h->ev_free.size() >= 2) { a = h->ev_free.back(); h->ev_free.pop_back(); b = h->ev_free.back(); h->ev_free.pop_back(); }
        else { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
        idx = h->ev_pool.size();
        h->ev_pool.push_back(a); h->ev_pool.push_back(b); h->ev_slot.push_back(slot);
        (void)hipEventRecord(a, h->stream);
    }
    void stop() { if (on) (void)hipEventRecord(h->ev_pool[idx + 1], h->stream); }
};

// where the QP kernel selected for this handle keeps its per-knot factors
upr_fb_src fb_source(const upr_batch* h) {
    upr_fb_src s{};   // (kind 0 = no source: the caller refuses to launch)
    const upr_dims& d = h->d;
    if (h->use_qp3) {
        auto fill = [&](auto cfg) {
            typedef decltype(cfg) C; typedef upr_qp3_ws<C> W; typedef upr_qp3_far<C> F;
            s.kind = 3; s.k_base = W::far + F::Ks; s.k_stride = C::NQ * C::NX; s.lji_base = 0; s.lji_stride = 0;
            s.lfi_base = W::far + F::lfi; s.lfi_stride = C::NLF; s.lsi_base = W::far + F::lsi; s.lsi_stride = C::NLS; s.lsi_sb = C::SB;
        };
        // (these offsets lie in front of everything that depends on the workgroup size or on ROWS / SOFT)
        if (h->use_qp3 == 3 && h->jit) {   // a run-time instantiated shape: the offsets its own info kernel reported
            const int* o = h->jit->fbo;
            s.kind = 3; s.k_base = o[0]; s.k_stride = o[1]; s.lji_base = 0; s.lji_stride = 0;
            s.lfi_base = o[2]; s.lfi_stride = o[3]; s.lsi_base = o[4]; s.lsi_stride = o[5]; s.lsi_sb = o[6];
        }
        else if (h->use_qp3 == 1) fill(upr_qp3_cfg<9, 1, 4, 3, 20, 256>());
#define X(a, b, c, e, n, rows, sf, dense) else if (qp3_match(h->P, h->d, a, b, c, e, n, rows, sf, dense)) fill(upr_qp3_cfg<a, b, c, e, n, 256, rows, sf, dense>());
        UPR_QP3_EXTRA(X)
#undef X
        return s;
    }
    if (h->use_qp2) {
#define X(a, b, c, e) if (h->P.nq == a && h->P.nb == b && h->P.nc == c && h->P.nf == e) { typedef upr_qp2_dims<a, b, c, e> D; upr_qp2_ws<D> w(d.N, d.neN); \
        s.kind = 2; s.k_base = w.store + D::SS_V; s.k_stride = D::SS_STRIDE; s.lji_base = w.store + D::SS_LJI; s.lji_stride = D::SS_STRIDE; \
        s.lfi_base = w.pre + D::PR_LFI; s.lfi_stride = D::PR_STRIDE; s.lsi_base = w.pre + D::PR_LSI; s.lsi_stride = D::PR_STRIDE; s.lsi_sb = d.ne; return s; }
        UPR_QP2_SHAPES(X)
#undef X
    }
    s.kind = 1; s.k_base = d.ws_store + d.ss_kx; s.k_stride = d.ss_stride; s.lji_base = d.ws_store + d.ss_hjj; s.lji_stride = d.ss_stride;
    s.lfi_base = d.ws_store + d.ss_hff; s.lfi_stride = d.ss_stride; s.lsi_base = d.ws_store + d.ss_sinv; s.lsi_stride = d.ss_stride; s.lsi_sb = d.ne;
    return s;
}

// Longest-first dispatch of the QP launch.  One workgroup solves one instance and a launch of B instances runs on
// 2 x 256 workgroup slots, so its duration is set by the slot that draws the largest SUM of IPM iteration counts: with
// the headline's 10..13 iterations per instance, arrival order gives 25 on some slot against a mean of 22.7.  The
// iteration count of an instance's previous QP predicts the next one well (same problem in a cold-start sweep, the
// neighbouring problem in closed loop), so the launch hands the instances out sorted by it, longest first (LPT rule):
// the short ones fill the gaps at the end.  The rank of an instance (ties by index) is computed by its own workgroup of the
// line-search launch that follows the QP (upr_linesearch.h, order_out): it only permutes which workgroup solves which instance.
int advance_impl(upr_batch* h) {
    const upr_dims& d = h->d;
    if (!h->guess_set) {
        hipLaunchKernelGGL(prepare_kernel, dim3(h->B), dim3(256), 0, h->stream, h->dP, d, h->B, h->has_prev ? 1 : 0, h->tprev,
                           h->xs_prev, h->us_prev, h->t0, h->x0, h->xs, h->us, h->stats, h->done);
        UPR_HIP(hipGetLastError());
    } else {
        UPR_HIP(hipMemsetAsync(h->stats, 0, sizeof(double) * h->B * UPR_NSTATS, h->stream));
        UPR_HIP(hipMemsetAsync(h->done, 0, sizeof(int) * h->B, h->stream));
        h->guess_set = false;
    }
    const int sqp_iters = h->sqp_iters_next > 0 ? h->sqp_iters_next : h->P.sqp_iters;
    h->sqp_iters_next = 0;
    for (int it = 0; it < sqp_iters; ++it) {
        { KernelTimer T(h, 0); if (do_linearize(h, traj_lin_args(h))) return 1; T.stop(); }
        {
            upr_qp_args Q = make_qp_args(h);
            // the production kernel writes the feedback gains of the advance's LAST QP itself (no gather kernel afterwards)
            if (h->fb && h->fb_fused && it == sqp_iters - 1) Q.fb = h->fb;
            if (h->order_on && h->order_valid) Q.order = h->order;
            KernelTimer T(h, 1); if (launch_qp(h, Q)) return 1; T.stop();
        }
        upr_ls_args L;
        L.P = h->dP; L.d = d; L.xs = h->xs; L.us = h->us; L.x0 = h->x0; L.t0 = h->t0; L.body_params = h->body_params;
        L.way_p = h->way_p; L.way_q = upr_has_orientation_cost(&h->P) ? h->way_q : nullptr; L.lin = h->lin; L.ws = h->ws; L.stats = h->stats; L.done = h->done; L.iter = it; L.dyn = h->dyn0; L.pflag = h->pflag;
        // folded into the line search's launch: the dispatch order of the next QP (rounds 2 - 3a: a one-workgroup counting sort
        // of its own) and, behind the advance's last line search, the copy of the solution the next warm start reads
        if (h->order_on) { L.order_out = h->order; L.iter_key = h->iter_key; }
        if (it == sqp_iters - 1) { L.xs_prev = h->xs_prev; L.us_prev = h->us_prev; L.tprev = h->tprev; }
        { KernelTimer T(h, 2); int rc = (h->P.nq == 6) ? launch_linesearch<6>(h, L) : launch_linesearch<9>(h, L); if (rc) return 1; T.stop(); }
        if (h->order_on) h->order_valid = true;
    }
    if (h->fb && !(h->fb_fused && sqp_iters > 0)) {   // sqp.use_feedback_policy: gains of the last QP, before anything overwrites its factors
        if (fb_source(h).kind == 0) return fail("feedback gains: the selected QP kernel names no source for them");
        if (d.ne <= 6 && d.nfc <= 12) hipLaunchKernelGGL((feedback_kernel<6, 12>), dim3(h->B), dim3(256), 0, h->stream, h->dP, d, fb_source(h), h->ws, h->lin, h->Df, h->stats, h->fb);
        else hipLaunchKernelGGL((feedback_kernel<6 * UPR_MAX_BODIES, 3 * UPR_MAX_CONTACTS>), dim3(h->B), dim3(256), 0, h->stream, h->dP, d, fb_source(h), h->ws, h->lin, h->Df, h->stats, h->fb);
        UPR_HIP(hipGetLastError());
    }
    h->hdyn_prev = h->hdyn0;
    // remember the solution for the next warm start / policy evaluation (the last line search did, unless there was none)
    if (sqp_iters <= 0) {
        UPR_HIP(hipMemcpyAsync(h->xs_prev, h->xs, sizeof(double) * h->B * (d.N + 1) * d.nx, hipMemcpyDeviceToDevice, h->stream));
        UPR_HIP(hipMemcpyAsync(h->us_prev, h->us, sizeof(double) * h->B * d.N * d.nu, hipMemcpyDeviceToDevice, h->stream));
        UPR_HIP(hipMemcpyAsync(h->tprev, h->t0, sizeof(double) * h->B, hipMemcpyDeviceToDevice, h->stream));
    }
    h->has_prev = true;
    return 0;
}

}  // namespace

extern "C" {

const char* upr_last_error(void) { return g_err.c_str(); }

int upr_device_available(void) {
    int n = 0;
    return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
}

int upr_core_object_dynamics(const upr_problem* P, const double* body_params, int n, const double* forces, const double* C,
                             const double* w, const double* al, const double* a, double* out) {
    if (need_device()) return 1;
    if (!P || P->nb < 1 || P->nb > UPR_MAX_BODIES || P->nc < 0 || P->nc > UPR_MAX_CONTACTS) return fail("bad problem dims");
    if (n <= 0) return 0;
    const int nfc = P->nf * P->nc, nb = P->nb;
    DevBuf<upr_problem> dP; DevBuf<double> dbp, df, dC, dw, dal, da, dout;
    if (dP.alloc(1) || dbp.alloc((size_t)nb * 10) || df.alloc((size_t)n * nfc + 1) || dC.alloc((size_t)n * 9) || dw.alloc((size_t)n * 3) ||
        dal.alloc((size_t)n * 3) || da.alloc((size_t)n * 3) || dout.alloc((size_t)n * 6 * nb)) return 1;
    UPR_HIP(hipMemcpy(dP, P, sizeof(upr_problem), hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dbp, body_params, sizeof(double) * nb * 10, hipMemcpyHostToDevice));
    if (nfc) UPR_HIP(hipMemcpy(df, forces, sizeof(double) * n * nfc, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dC, C, sizeof(double) * n * 9, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dw, w, sizeof(double) * n * 3, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dal, al, sizeof(double) * n * 3, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(da, a, sizeof(double) * n * 3, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(core_object_dynamics_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, dP.p, dbp.p, n, df.p, dC.p, dw.p, dal.p, da.p, dout.p);
    UPR_HIP(hipGetLastError());
    UPR_HIP(hipMemcpy(out, dout, sizeof(double) * n * 6 * nb, hipMemcpyDeviceToHost));
    return 0;
}

int upr_core_friction_rows(const upr_problem* P, int n, const double* forces, double* out) {
    if (need_device()) return 1;
    if (!P || P->nc < 1 || P->nc > UPR_MAX_CONTACTS) return fail("bad problem dims");
    if (n <= 0) return 0;
    DevBuf<upr_problem> dP; DevBuf<double> df, dout;
    if (dP.alloc(1) || df.alloc((size_t)n * 3 * P->nc) || dout.alloc((size_t)n * 5 * P->nc)) return 1;
    UPR_HIP(hipMemcpy(dP, P, sizeof(upr_problem), hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(df, forces, sizeof(double) * n * 3 * P->nc, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(core_friction_rows_kernel, dim3((n * P->nc + 63) / 64), dim3(64), 0, 0, dP.p, n, df.p, dout.p);
    UPR_HIP(hipGetLastError());
    UPR_HIP(hipMemcpy(out, dout, sizeof(double) * n * 5 * P->nc, hipMemcpyDeviceToHost));
    return 0;
}

upr_batch* upr_batch_create(const upr_problem* P, int B, const double* body_params, const double* way_p) {
    if (check_problem(P)) return nullptr;
    if (B < 1) { fail("B must be >= 1"); return nullptr; }
    if (need_device()) return nullptr;
    upr_batch* h = new upr_batch();
    if (hipGetDevice(&h->device) != hipSuccess) { fail("hipGetDevice failed"); delete h; return nullptr; }
    h->P = *P; h->d = upr_make_dims(P); h->B = B;
    h->nxf = h->d.nx + 9 * P->n_dyn;
    h->hdyn0.assign((size_t)B * 9 * P->n_dyn, 0.0); h->hdyn_prev = h->hdyn0; h->htprev.assign(B, 0.0);
    const upr_dims& d = h->d;
    if (d.nx > UPR_LPK) { fail("nx exceeds the 32 tangent lanes of the linearisation kernel"); delete h; return nullptr; }
    // collision rows (state-polytopic inequalities) are implemented in the generic kernel only
    const bool soft = P->soft_state_box || P->soft_input_box || P->soft_poly || P->soft_eq;
    const bool plain = P->n_pairs + P->n_proj == 0 && !soft;   // rows only the generic kernel has
    h->use_qp3 = qp3_variant(*P, h->d);
    h->use_qp2 = qp2_has_shape(*P) && plain;
    const bool forced_other = (getenv("UPR_QP_KERNEL") && atoi(getenv("UPR_QP_KERNEL")) < 3) || (getenv("UPR_QP_GENERIC") && atoi(getenv("UPR_QP_GENERIC")) != 0);
    if (getenv("UPR_QP3_JIT") && atoi(getenv("UPR_QP3_JIT")) == 2) h->use_qp3 = 0;   // (diagnostic: the run-time instantiation also for the listed shapes)
    if (h->use_qp3 == 0 && !forced_other && jit_wanted(B) && jit_capable(*P, h->d)) {
        // a shape outside the list: instantiate the production structure for it now (falls back with a note if that fails)
        upr_jit_kernel* k = nullptr;
        if (jit_get(*P, h->d, &k) == 0) { h->use_qp3 = 3; h->jit = k; }
        else fprintf(stderr, "libupright_mi: %s -- using the %s kernel instead\n", g_err.c_str(), h->use_qp2 ? "second-structure" : "generic");
    }
    // UPR_QP_KERNEL = 1 (generic) | 2 | 3 selects an older structure for A/B measurements and tests
    if (const char* e = getenv("UPR_QP_KERNEL")) { int v = atoi(e); if (v < 3) h->use_qp3 = 0; if (v < 2) h->use_qp2 = false; }
    if (const char* e = getenv("UPR_QP_GENERIC")) { if (atoi(e) != 0) { h->use_qp2 = false; h->use_qp3 = 0; } }
    h->qp_nt = h->use_qp3 ? 256 : 128;
    if (const char* e = getenv("UPR_QP_NT")) h->qp_nt = atoi(e);
    if (h->use_qp3 == 2) h->qp_nt = 256;
    h->fb_fused = h->use_qp3 != 0;
    if (const char* e = getenv("UPR_FB_FUSED")) h->fb_fused = h->fb_fused && atoi(e) != 0;
    {   // every QP kernel indexes the instance workspace with the same stride: the largest any selectable one needs
        size_t need = qp2_ws_doubles(*P, h->d);
        if ((size_t)h->d.ws_stride < need) h->d.ws_stride = (int)need;
        if (h->use_qp3) { need = (h->use_qp3 == 3) ? h->jit->ws : qp3_ws_doubles(*P, h->d, h->use_qp3); if ((size_t)h->d.ws_stride < need) h->d.ws_stride = (int)need; }
    }
    {
        char buf[128];
        if (h->use_qp3 == 1) snprintf(buf, sizeof(buf), "upr_qp3_kernel<upr_qp3_cfg<9, 1, 4, 3, 20, %d, %s, false, false>>", h->qp_nt, h->d.no > 0 ? "true" : "false");
#define X(a, b, c, e, n, rows, sf, dense) else if (h->use_qp3 == 2 && qp3_match(*P, h->d, a, b, c, e, n, rows, sf, dense)) snprintf(buf, sizeof(buf), "upr_qp3_kernel<upr_qp3_cfg<%d, %d, %d, %d, %d, 256, %s, %s, %s>>", a, b, c, e, n, #rows, #sf, #dense);
        UPR_QP3_EXTRA(X)
#undef X
        else if (h->use_qp3 == 3) snprintf(buf, sizeof(buf), "upr_qp3_jit<upr_qp3_cfg<%s>>", h->jit->cfg.c_str());
        else if (h->use_qp2) snprintf(buf, sizeof(buf), "upr_qp2_kernel<upr_qp2_dims<%d, %d, %d, %d>, %d>", P->nq, P->nb, P->nc, P->nf, h->qp_nt == 512 ? 128 : h->qp_nt);
        else snprintf(buf, sizeof(buf), "upr_qp_kernel<%d>", generic_nt(h));
        h->qp_name = buf;
    }
    if (const char* e = getenv("UPR_LIN_MFMA")) h->use_mfma = atoi(e) != 0;
    if (const char* e = getenv("UPR_LIN2")) h->lin2 = atoi(e) != 0;
    if (const char* e = getenv("UPR_QP_ORDER")) h->order_on = atoi(e) != 0;
    auto bad = [&]() { upr_batch_destroy(h); return (upr_batch*)nullptr; };
    if (hipStreamCreate(&h->stream) != hipSuccess) { fail("hipStreamCreate failed"); return bad(); }
    if (hipMalloc((void**)&h->dP, sizeof(upr_problem)) != hipSuccess) { fail("hipMalloc failed"); return bad(); }
    hipMemcpy(h->dP, P, sizeof(upr_problem), hipMemcpyHostToDevice);
    const size_t n1 = d.N + 1;
    if (dev_alloc(&h->body_params, (size_t)B * d.nb * 10) || dev_alloc(&h->way_p, (size_t)B * P->n_way * 3) || dev_alloc(&h->way_q, (size_t)B * P->n_way * 4) ||
        // (observation: time, state and the dynamic obstacles' states in ONE block, [t B][x B nx][dyn B 9 n_dyn] -- a control period
        //  copies it in one piece, upr_batch_tick)
        dev_alloc(&h->t0, (((size_t)B + 1) & ~(size_t)1) + (size_t)B * d.nx + (size_t)B * 9 * P->n_dyn) || dev_alloc(&h->xs, (size_t)B * n1 * d.nx) || dev_alloc(&h->us, (size_t)B * d.N * d.nu) ||
        dev_alloc(&h->xs_prev, (size_t)B * n1 * d.nx) || dev_alloc(&h->us_prev, (size_t)B * d.N * d.nu) || dev_alloc(&h->tprev, B) ||
        dev_alloc(&h->lin, (size_t)B * n1 * d.lin_stride) || dev_alloc(&h->Df, (size_t)B * d.ne * d.nfc) ||
        dev_alloc(&h->ws, (size_t)B * d.ws_stride) || dev_alloc(&h->stats, (size_t)B * UPR_NSTATS) || dev_alloc(&h->done, B) || dev_alloc(&h->order, B) || dev_alloc(&h->iter_key, ((size_t)B + 3) & ~(size_t)3) ||
        (P->use_feedback_policy && dev_alloc(&h->fb, (size_t)B * d.N * d.nu * d.nx)) ||
        (P->n_dyn && dev_alloc(&h->pflag, B)))
        return bad();
    h->x0 = h->t0 + (((size_t)B + 1) & ~(size_t)1);   // (the states stay 16-byte aligned)
    if (P->n_dyn) h->dyn0 = h->x0 + (size_t)B * d.nx;
    hipMemcpy(h->body_params, body_params, sizeof(double) * B * d.nb * 10, hipMemcpyHostToDevice);
    hipMemcpy(h->way_p, way_p, sizeof(double) * B * P->n_way * 3, hipMemcpyHostToDevice);
    {   // target orientations default to the identity quaternion (xyzw) until upr_batch_set_target_orientations
        std::vector<double> qi((size_t)B * P->n_way * 4, 0.0);
        for (size_t i = 3; i < qi.size(); i += 4) qi[i] = 1.0;
        hipMemcpy(h->way_q, qi.data(), sizeof(double) * qi.size(), hipMemcpyHostToDevice);
    }
    // constant d(object_dynamics)/d(forces): unit forces through the wrench map (contact_constraints.h:107-157),
    // divided by the body mass and sqrt(6 nb) (balancing_constraints.cpp:144-151), sign of (GIF - F)
    h->hDf.assign((size_t)B * d.ne * d.nfc, 0.0);
    std::vector<double> unit(d.nfc), Fw(6 * d.nb);
    const double scale = 1.0 / std::sqrt(6.0 * d.nb);
    for (int b = 0; b < B; ++b) {
        const double* bp = body_params + (size_t)b * d.nb * 10;
        for (int j = 0; j < d.nfc; ++j) {
            std::fill(unit.begin(), unit.end(), 0.0);
            unit[j] = 1.0;
            upr_object_wrenches(P, bp, unit.data(), Fw.data());
            for (int bb = 0; bb < d.nb; ++bb)
                for (int r = 0; r < 6; ++r) h->hDf[((size_t)b * d.ne + 6 * bb + r) * d.nfc + j] = -scale * Fw[6 * bb + r] / bp[10 * bb];
        }
    }
    hipMemcpy(h->Df, h->hDf.data(), sizeof(double) * h->hDf.size(), hipMemcpyHostToDevice);
    if (hipDeviceSynchronize() != hipSuccess) { fail("device synchronisation failed in create"); return bad(); }
    return h;
}

void upr_batch_destroy(upr_batch* h) {
    if (!h) return;
    hipFree(h->dP); hipFree(h->body_params); hipFree(h->way_p); hipFree(h->way_q); hipFree(h->t0) /* (x0 and dyn0 live in the same block) */; hipFree(h->xs); hipFree(h->us);
    if (h->fb) hipFree(h->fb);
    if (h->pflag) hipFree(h->pflag);
    hipFree(h->xs_prev); hipFree(h->us_prev); hipFree(h->tprev); hipFree(h->lin); hipFree(h->Df); hipFree(h->ws); hipFree(h->stats);
    hipFree(h->done); hipFree(h->order); hipFree(h->iter_key); if (h->pin) (void)hipHostFree(h->pin); if (h->tick_exec) (void)hipGraphExecDestroy(h->tick_exec); hipFree(h->prof); hipFree(h->kkt);
    hipFree(h->ev_t); hipFree(h->ev_xo); hipFree(h->ev_x) /* (ev_u: same block) */;
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int upr_batch_reset(upr_batch* h, const double* way_p) {
    UPR_ENTER(h);
    if (way_p) UPR_HIP(hipMemcpyAsync(h->way_p, way_p, sizeof(double) * h->B * h->P.n_way * 3, hipMemcpyHostToDevice, h->stream));
    h->has_prev = false;
    UPR_HIP(hipStreamSynchronize(h->stream));
    h->guess_set = false;
    return 0;
}

int upr_batch_set_target_orientations(upr_batch* h, const double* way_q) {
    UPR_ENTER(h);
    if (!way_q) return fail("upr_batch_set_target_orientations: way_q is NULL");
    std::vector<double> q(way_q, way_q + (size_t)h->B * h->P.n_way * 4);
    for (size_t i = 0; i < q.size(); i += 4) {   // unit quaternions (the reference builds Quatd from the target's coefficients)
        const double n = std::sqrt(q[i] * q[i] + q[i + 1] * q[i + 1] + q[i + 2] * q[i + 2] + q[i + 3] * q[i + 3]);
        if (!(n > 0)) return fail("upr_batch_set_target_orientations: zero quaternion");
        for (int c = 0; c < 4; ++c) q[i + c] /= n;
    }
    UPR_HIP(hipMemcpyAsync(h->way_q, q.data(), sizeof(double) * q.size(), hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static int set_observation_core(upr_batch* h, const double* t, int t_stride, const double* x) {
    UPR_ENTER(h);
    std::vector<double> tt(h->B);
    for (int b = 0; b < h->B; ++b) tt[b] = t[(size_t)b * (t_stride ? 1 : 0)];
    UPR_HIP(hipMemcpyAsync(h->t0, tt.data(), sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipMemcpyAsync(h->x0, x, sizeof(double) * h->B * h->d.nx, hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static int set_guess_core(upr_batch* h, const double* xs, const double* us) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    UPR_HIP(hipMemcpyAsync(h->xs, xs, sizeof(double) * h->B * (d.N + 1) * d.nx, hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipMemcpyAsync(h->us, us, sizeof(double) * h->B * d.N * d.nu, hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipStreamSynchronize(h->stream));
    h->guess_set = true;
    return 0;
}

int upr_batch_set_sqp_iterations(upr_batch* h, int n) {
    UPR_ENTER(h);
    if (n < 0 || n > 1000) return fail("upr_batch_set_sqp_iterations: n out of range");
    h->sqp_iters_next = n;
    return 0;
}

int upr_batch_advance_async(upr_batch* h) {
    UPR_ENTER(h);
    return advance_impl(h);
}

int upr_batch_sync(upr_batch* h) {
    UPR_ENTER(h);
    UPR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int upr_batch_advance(upr_batch* h) {
    UPR_ENTER(h);
    auto t0 = std::chrono::steady_clock::now();
    if (advance_impl(h)) return 1;
    UPR_HIP(hipStreamSynchronize(h->stream));
    h->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

static int get_solution_core(upr_batch* h, double* ts, double* xs, double* us) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    UPR_HIP(hipStreamSynchronize(h->stream));
    if (ts) {
        std::vector<double> t0(h->B);
        UPR_HIP(hipMemcpy(t0.data(), h->t0, sizeof(double) * h->B, hipMemcpyDeviceToHost));
        for (int b = 0; b < h->B; ++b) for (int k = 0; k <= d.N; ++k) ts[(size_t)b * (d.N + 1) + k] = t0[b] + k * h->P.dt;
    }
    if (xs) UPR_HIP(hipMemcpy(xs, h->xs, sizeof(double) * h->B * (d.N + 1) * d.nx, hipMemcpyDeviceToHost));
    if (us) UPR_HIP(hipMemcpy(us, h->us, sizeof(double) * h->B * d.N * d.nu, hipMemcpyDeviceToHost));
    return 0;
}

static int evaluate_core(upr_batch* h, const double* t, int t_stride, double* x_out, double* u_out) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    // (per-tick path of the closed loop: scratch preallocated in the handle on first use)
    if (!h->ev_t && (dev_alloc(&h->ev_t, h->B) || dev_alloc(&h->ev_xo, (size_t)h->B * d.nx) || dev_alloc(&h->ev_x, (size_t)h->B * (d.nx + d.nu)))) return 1;
    h->ev_u = h->ev_x + (size_t)h->B * d.nx;   // (one block: a control period copies both out in one piece)
    std::vector<double> tt(h->B);
    for (int b = 0; b < h->B; ++b) tt[b] = t[(size_t)b * (t_stride ? 1 : 0)];
    UPR_HIP(hipMemcpyAsync(h->ev_t, tt.data(), sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(evaluate_kernel, dim3(h->B), dim3(64), 0, h->stream, h->dP, d, h->B, h->tprev, h->xs_prev, h->us_prev, h->ev_t, 1, h->ev_x, h->ev_u);
    UPR_HIP(hipGetLastError());
    UPR_HIP(hipMemcpyAsync(x_out, h->ev_x, sizeof(double) * h->B * d.nx, hipMemcpyDeviceToHost, h->stream));
    UPR_HIP(hipMemcpyAsync(u_out, h->ev_u, sizeof(double) * h->B * d.nu, hipMemcpyDeviceToHost, h->stream));
    UPR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static int evaluate_policy_core(upr_batch* h, const double* t, int t_stride, const double* x_obs, double* x_out, double* u_out) {
    UPR_ENTER(h);
    if (!h->fb) return fail("upr_batch_evaluate_policy: the batch was created with use_feedback_policy = 0");
    const upr_dims& d = h->d;
    if (!h->ev_t && (dev_alloc(&h->ev_t, h->B) || dev_alloc(&h->ev_xo, (size_t)h->B * d.nx) || dev_alloc(&h->ev_x, (size_t)h->B * (d.nx + d.nu)))) return 1;
    h->ev_u = h->ev_x + (size_t)h->B * d.nx;   // (one block: a control period copies both out in one piece)
    std::vector<double> tt(h->B);
    for (int b = 0; b < h->B; ++b) tt[b] = t[(size_t)b * (t_stride ? 1 : 0)];
    UPR_HIP(hipMemcpyAsync(h->ev_t, tt.data(), sizeof(double) * h->B, hipMemcpyHostToDevice, h->stream));
    UPR_HIP(hipMemcpyAsync(h->ev_xo, x_obs, sizeof(double) * h->B * d.nx, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(evaluate_policy_kernel, dim3(h->B), dim3(64), 0, h->stream, h->dP, d, h->B, h->tprev, h->xs_prev, h->us_prev, h->fb, h->ev_t, h->ev_xo, h->ev_x, h->ev_u);
    UPR_HIP(hipGetLastError());
    UPR_HIP(hipMemcpyAsync(x_out, h->ev_x, sizeof(double) * h->B * d.nx, hipMemcpyDeviceToHost, h->stream));
    UPR_HIP(hipMemcpyAsync(u_out, h->ev_u, sizeof(double) * h->B * d.nu, hipMemcpyDeviceToHost, h->stream));
    UPR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static int get_feedback_core(upr_batch* h, double* K) {
    UPR_ENTER(h);
    if (!h->fb) return fail("upr_batch_get_feedback: the batch was created with use_feedback_policy = 0");
    const upr_dims& d = h->d;
    UPR_HIP(hipStreamSynchronize(h->stream));
    UPR_HIP(hipMemcpy(K, h->fb, sizeof(double) * h->B * d.N * d.nu * d.nx, hipMemcpyDeviceToHost));
    return 0;
}

double upr_batch_last_solve_ms(const upr_batch* h) { return h ? h->last_ms : 0.0; }

const char* upr_batch_qp_kernel_name(const upr_batch* h) { return h ? h->qp_name.c_str() : ""; }

/* restore == 0: keep a copy of the per-instance statistics and of the QP dispatch keys of the last advance; restore != 0: put that
 * copy back.  Brackets a query that solves one more QP on the handle (upr_batch_qp_kkt behind valueFunction): afterwards
 * upr_batch_get_stats and the longest-first dispatch order describe the SOLVE again, not the query's QP. */
int upr_batch_hold_stats(upr_batch* h, int restore) {
    UPR_ENTER(h);
    const size_t ns = sizeof(double) * h->B * UPR_NSTATS, nk = ((size_t)h->B + 3) & ~(size_t)3;
    UPR_HIP(hipStreamSynchronize(h->stream));
    if (!restore) {
        h->held_stats.resize((size_t)h->B * UPR_NSTATS); h->held_keys.resize(nk);
        UPR_HIP(hipMemcpy(h->held_stats.data(), h->stats, ns, hipMemcpyDeviceToHost));
        UPR_HIP(hipMemcpy(h->held_keys.data(), h->iter_key, nk, hipMemcpyDeviceToHost));
        return 0;
    }
    if (h->held_stats.empty()) return fail("upr_batch_hold_stats: nothing held on this handle");
    UPR_HIP(hipMemcpy(h->stats, h->held_stats.data(), ns, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(h->iter_key, h->held_keys.data(), nk, hipMemcpyHostToDevice));
    h->held_stats.clear(); h->held_keys.clear();
    return 0;
}

int upr_batch_get_stats(upr_batch* h, double* stats) {
    UPR_ENTER(h);
    UPR_HIP(hipStreamSynchronize(h->stream));
    UPR_HIP(hipMemcpy(stats, h->stats, sizeof(double) * h->B * UPR_NSTATS, hipMemcpyDeviceToHost));
    return 0;
}

// points mode of the linearisation kernel: records of n arbitrary (x, u) pairs back on the host
static int linearize_points_impl(upr_batch* h, int n, const int* inst, const double* t, const double* x, const double* u,
                                 std::vector<double>& rec, double* ee, const double* dyn_pts = nullptr) {
    const upr_dims& d = h->d;
    for (int i = 0; i < n; ++i) if (inst[i] < 0 || inst[i] >= h->B) return fail("instance index out of range");
    DevBuf<int> dinst; DevBuf<double> dt_, dx, du, dlin, dee, ddyn;
    if (dinst.alloc(n) || dt_.alloc(n) || dx.alloc((size_t)n * d.nx) || du.alloc((size_t)n * d.nu) ||
        dlin.alloc((size_t)n * d.lin_stride) || dee.alloc((size_t)n * 3)) return 1;
    UPR_HIP(hipMemcpy(dinst, inst, sizeof(int) * n, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dt_, t, sizeof(double) * n, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(dx, x, sizeof(double) * n * d.nx, hipMemcpyHostToDevice));
    UPR_HIP(hipMemcpy(du, u, sizeof(double) * n * d.nu, hipMemcpyHostToDevice));
    upr_lin_args A;
    A.P = h->dP; A.d = d; A.body_params = h->body_params; A.way_p = h->way_p; A.way_q = upr_has_orientation_cost(&h->P) ? h->way_q : nullptr; A.t0 = dt_; A.xs = dx; A.us = du; A.inst = dinst;
    A.lin = dlin; A.ee_out = dee; A.npoints = n; A.Df = h->Df;
    if (h->P.n_dyn) {   // points mode: the obstacle state of every point as given
        if (ddyn.alloc((size_t)n * 9 * h->P.n_dyn)) return 1;
        if (dyn_pts) UPR_HIP(hipMemcpy(ddyn, dyn_pts, sizeof(double) * n * 9 * h->P.n_dyn, hipMemcpyHostToDevice));
        A.dyn = ddyn; A.pflag = h->pflag;
    }
    if (do_linearize(h, A)) return 1;
    UPR_HIP(hipStreamSynchronize(h->stream));
    rec.assign((size_t)n * d.lin_stride, 0.0);
    UPR_HIP(hipMemcpy(rec.data(), dlin, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
    if (ee) UPR_HIP(hipMemcpy(ee, dee, sizeof(double) * n * 3, hipMemcpyDeviceToHost));
    return 0;
}

int upr_batch_linearize_points(upr_batch* h, int n, const int* inst, const double* t, const double* x, const double* u,
                               double* g, double* gx, double* cost, double* grad, double* hess, double* ee) {
    UPR_ENTER(h);
    if (n <= 0) return 0;
    const upr_dims& d = h->d;
    std::vector<double> rec, xr, dyn;
    if (h->P.n_dyn) {   // interface states [robot x, obstacle]; Jacobian outputs keep the 3 nq robot columns
        const size_t nd9 = 9 * (size_t)h->P.n_dyn;
        xr.resize((size_t)n * d.nx); dyn.resize((size_t)n * nd9);
        for (int i = 0; i < n; ++i) {
            std::memcpy(xr.data() + (size_t)i * d.nx, x + (size_t)i * h->nxf, sizeof(double) * d.nx);
            std::memcpy(dyn.data() + (size_t)i * nd9, x + (size_t)i * h->nxf + d.nx, sizeof(double) * nd9);
        }
        x = xr.data();
    }
    if (linearize_points_impl(h, n, inst, t, x, u, rec, ee, h->P.n_dyn ? dyn.data() : nullptr)) return 1;
    for (int i = 0; i < n; ++i) {
        const double* r = rec.data() + (size_t)i * d.lin_stride;
        if (g) std::memcpy(g + (size_t)i * d.ne, r + d.lin_g, sizeof(double) * d.ne);
        if (gx) std::memcpy(gx + (size_t)i * d.ne * d.nx, r + d.lin_gx, sizeof(double) * d.ne * d.nx);
        if (cost) cost[i] = r[d.lin_cost];
        if (grad) std::memcpy(grad + (size_t)i * d.nq, r + d.lin_grad, sizeof(double) * d.nq);
        if (hess) for (int a = 0; a < d.nq; ++a) for (int c = 0; c < d.nq; ++c) hess[((size_t)i * d.nq + a) * d.nq + c] = r[d.lin_hess + upr_tri(d.nq, a, c)];
    }
    return 0;
}

int upr_batch_obstacle_rows(upr_batch* h, int n, const double* x, double* dd, double* dq) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    if (d.no == 0) return fail("upr_batch_obstacle_rows: the problem has no collision pairs / projectile rows");
    if (n <= 0) return 0;
    std::vector<int> inst(n, 0);
    std::vector<double> t(n, 0.0), u((size_t)n * d.nu, 0.0), rec, xr((size_t)n * d.nx), dyn((size_t)n * 9 * h->P.n_dyn);
    for (int i = 0; i < n; ++i) {   // interface states: [robot x, obstacle r v a]
        std::memcpy(xr.data() + (size_t)i * d.nx, x + (size_t)i * h->nxf, sizeof(double) * d.nx);
        if (h->P.n_dyn) std::memcpy(dyn.data() + (size_t)i * 9 * h->P.n_dyn, x + (size_t)i * h->nxf + d.nx, sizeof(double) * 9 * h->P.n_dyn);
    }
    if (linearize_points_impl(h, n, inst.data(), t.data(), xr.data(), u.data(), rec, nullptr, h->P.n_dyn ? dyn.data() : nullptr)) return 1;
    for (int i = 0; i < n; ++i) {
        const double* r = rec.data() + (size_t)i * d.lin_stride + d.lin_obs;
        std::memcpy(dd + (size_t)i * d.no, r, sizeof(double) * d.no);
        if (dq) std::memcpy(dq + (size_t)i * d.no * d.nq, r + d.no, sizeof(double) * d.no * d.nq);
    }
    return 0;
}

int upr_batch_eq_input_jacobian(upr_batch* h, int inst, double* gu) {
    UPR_ENTER(h);
    if (inst < 0 || inst >= h->B) return fail("instance index out of range");
    const upr_dims& d = h->d;
    for (int r = 0; r < d.ne; ++r) {
        for (int j = 0; j < d.nq; ++j) gu[(size_t)r * d.nu + j] = 0.0;
        for (int j = 0; j < d.nfc; ++j) gu[(size_t)r * d.nu + d.nq + j] = h->hDf[((size_t)inst * d.ne + r) * d.nfc + j];
    }
    return 0;
}

static int qp_step_core(upr_batch* h, double* dxs, double* dus) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    if (do_linearize(h, traj_lin_args(h))) return 1;
    if (launch_qp(h, make_qp_args(h))) return 1;
    UPR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> ws((size_t)h->B * d.ws_stride);
    UPR_HIP(hipMemcpy(ws.data(), h->ws, sizeof(double) * ws.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; ++b) {
        const double* w = ws.data() + (size_t)b * d.ws_stride;
        if (dxs) std::memcpy(dxs + (size_t)b * (d.N + 1) * d.nx, w + d.ws_dx, sizeof(double) * (d.N + 1) * d.nx);
        if (dus) std::memcpy(dus + (size_t)b * d.N * d.nu, w + d.ws_du, sizeof(double) * d.N * d.nu);
    }
    return 0;
}

/* One QP at the current trajectory with everything an independent KKT check needs: the step and the multipliers the
 * kernel ended with.  pi[B][N+1][nx] costates (pi_0 unused), nu[B][N][ne] stage-equality multipliers, yN[B][neN]
 * terminal-equality multipliers, lam[B][N+1][ni] inequality multipliers, ni = 2 nx + 2 nu + np + no in the slot order
 * [x lower][x upper][u lower][u upper][friction rows][collision / projectile rows]; *ni_out = ni.  Any pointer may be NULL. */
int upr_batch_qp_kkt(upr_batch* h, double* dxs, double* dus, double* pi, double* nu, double* yN, double* lam, int* ni_out) {
    UPR_ENTER(h);
    if (h->P.n_dyn) return fail("upr_batch_qp_kkt: not available with a dynamic obstacle (interface states)");
    const upr_dims& d = h->d;
    if (ni_out) *ni_out = d.ni_stage;
    const int kdoubles = upr_kkt_doubles(d);
    if (h->use_qp3 && !h->kkt && dev_alloc(&h->kkt, (size_t)h->B * kdoubles)) return 1;
    if (do_linearize(h, traj_lin_args(h))) return 1;
    upr_qp_args A = make_qp_args(h);
    if (h->use_qp3) { A.kkt = h->kkt; A.kkt_stride = kdoubles; }
    if (launch_qp(h, A)) return 1;
    UPR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> ws((size_t)h->B * d.ws_stride), kk;
    UPR_HIP(hipMemcpy(ws.data(), h->ws, sizeof(double) * ws.size(), hipMemcpyDeviceToHost));
    if (h->use_qp3) { kk.resize((size_t)h->B * kdoubles); UPR_HIP(hipMemcpy(kk.data(), h->kkt, sizeof(double) * kk.size(), hipMemcpyDeviceToHost)); }
    const int n1 = d.N + 1;
    int o_pi = d.ws_pi, o_nu = d.ws_nu, o_y = d.ws_yN, o_lam = d.ws_lam, o_t = d.ws_t;
    if (!h->use_qp3 && h->use_qp2) {
#define X(a, b, c, e) if (h->P.nq == a && h->P.nb == b && h->P.nc == c && h->P.nf == e) { upr_qp2_ws<upr_qp2_dims<a, b, c, e>> w(d.N, d.neN); o_pi = w.pi; o_nu = w.nu; o_y = w.yN; o_lam = w.lam; o_t = w.t; }
        UPR_QP2_SHAPES(X)
#undef X
    }
    for (int b = 0; b < h->B; ++b) {
        const double* w = ws.data() + (size_t)b * d.ws_stride;
        if (dxs) std::memcpy(dxs + (size_t)b * n1 * d.nx, w + d.ws_dx, sizeof(double) * n1 * d.nx);
        if (dus) std::memcpy(dus + (size_t)b * d.N * d.nu, w + d.ws_du, sizeof(double) * d.N * d.nu);
        const double *spi, *snu, *sy, *sl, *st;
        if (h->use_qp3) { const double* k = kk.data() + (size_t)b * kdoubles; spi = k; snu = spi + n1 * d.nx; sy = snu + d.N * d.ne; sl = sy + d.neN; st = sl + (size_t)n1 * d.ni_stage; }
        else { spi = w + o_pi; snu = w + o_nu; sy = w + o_y; sl = w + o_lam; st = w + o_t; }
        if (b == 0) h->kkt_slack.assign((size_t)h->B * n1 * d.ni_stage, 1.0);
        for (int k = 0; k < n1; ++k) for (int j = 0; j < d.ni_stage; ++j)
            if (upr_ineq_active(d, k, j)) h->kkt_slack[((size_t)b * n1 + k) * d.ni_stage + j] = st[(size_t)k * d.ni_stage + j];
        if (pi) std::memcpy(pi + (size_t)b * n1 * d.nx, spi, sizeof(double) * n1 * d.nx);
        if (nu) std::memcpy(nu + (size_t)b * d.N * d.ne, snu, sizeof(double) * d.N * d.ne);
        if (yN && d.neN) std::memcpy(yN + (size_t)b * d.neN, sy, sizeof(double) * d.neN);
        if (lam) {
            // the generic kernels keep a multiplier value in slots that are not rows of the stage (x rows of knot 0, u rows of
            // knot N, collision rows outside knots 1 .. N-1): report 0 there
            for (int k = 0; k < n1; ++k) for (int j = 0; j < d.ni_stage; ++j)
                lam[((size_t)b * n1 + k) * d.ni_stage + j] = upr_ineq_active(d, k, j) ? sl[(size_t)k * d.ni_stage + j] : 0.0;
        }
    }
    return 0;
}

/* Slacks t of the inequality rows at the exit of the QP the last upr_batch_qp_kkt call solved, t[B][N+1][ni] in the slot order of its
 * lam (1 where a slot is not a row of the knot): lam / t are the barrier weights of the last interior-point iterate. */
int upr_batch_qp_slacks(upr_batch* h, double* t) {
    UPR_ENTER(h);
    if (h->kkt_slack.empty()) return fail("upr_batch_qp_slacks: no upr_batch_qp_kkt call on this handle yet");
    if (t) std::memcpy(t, h->kkt_slack.data(), sizeof(double) * h->kkt_slack.size());
    return 0;
}

int upr_batch_device_ptrs(upr_batch* h, void** xs, void** us) {
    UPR_ENTER(h);
    if (xs) *xs = h->xs;
    if (us) *us = h->us;
    return 0;
}

/* debug: per-phase cycle counters of the production QP kernel, prof[B][4 waves][16]; allocate on first use.  The stamps are compiled
 * into run-time instantiations only (UPR_QP3_JIT=2 UPR_JIT_FLAGS="-DUPR_QP3_PROF ..."): the library's own kernels carry none. */
int upr_batch_qp_profile(upr_batch* h, double* out) {
    UPR_ENTER(h);
    {
        const char* f = getenv("UPR_JIT_FLAGS");
        if (h->use_qp3 != 3 || !f || !strstr(f, "-DUPR_QP3_PROF"))
            return fail("upr_batch_qp_profile: the phase stamps exist in run-time instantiations only: create the handle with UPR_QP3_JIT=2 UPR_JIT_FLAGS=-DUPR_QP3_PROF");
    }
    if (!h->prof) { if (dev_alloc(&h->prof, (size_t)h->B * 64)) return 1; return 0; }
    UPR_HIP(hipStreamSynchronize(h->stream));
    if (out) UPR_HIP(hipMemcpy(out, h->prof, sizeof(double) * h->B * 64, hipMemcpyDeviceToHost));
    UPR_HIP(hipMemset(h->prof, 0, sizeof(double) * h->B * 64));
    return 0;
}

/* debug / test accessor: the per-knot linearisation records of the current trajectory, lin[B][N+1][stride] */
int upr_batch_get_lin(upr_batch* h, double* lin, int* stride) {
    UPR_ENTER(h);
    UPR_HIP(hipStreamSynchronize(h->stream));
    if (stride) *stride = h->d.lin_stride;
    if (lin) UPR_HIP(hipMemcpy(lin, h->lin, sizeof(double) * h->B * (h->d.N + 1) * h->d.lin_stride, hipMemcpyDeviceToHost));
    return 0;
}

int upr_batch_enable_timing(upr_batch* h, int on) {
    UPR_ENTER(h);
    h->timing = (on == 2 || on == 3) ? on : (on != 0 ? 1 : 0);
    h->timing_qp_count = 0;
    if (h->timing) while (h->ev_free.size() < 512) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; h->ev_free.push_back(e); }
    for (int i = 0; i < 3; ++i) { h->k_ms[i] = 0; h->k_launches[i] = 0; }
    return 0;
}

int upr_batch_kernel_times(upr_batch* h, double* ms, int* launches) {
    UPR_ENTER(h);
    UPR_HIP(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->ev_slot.size(); ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]) == hipSuccess) { h->k_ms[h->ev_slot[i]] += t; h->k_launches[h->ev_slot[i]] += 1; }
        h->ev_free.push_back(h->ev_pool[2 * i]); h->ev_free.push_back(h->ev_pool[2 * i + 1]);
    }
    h->ev_pool.clear(); h->ev_slot.clear();
    for (int i = 0; i < 3; ++i) {
        ms[i] = h->k_launches[i] ? h->k_ms[i] / h->k_launches[i] : 0.0;
        if (launches) launches[i] = h->k_launches[i];
    }
    return 0;
}

/* copy the current solution into caller-owned DEVICE buffers (e.g. torch tensors handed to an RCCL
 * all-gather): xs_dst[B][N+1][nx], us_dst[B][N][nu]; asynchronous on the engine's stream. */
int upr_batch_copy_solution_device(upr_batch* h, void* xs_dst, void* us_dst) {
    UPR_ENTER(h);
    const upr_dims& d = h->d;
    if (xs_dst) UPR_HIP(hipMemcpyAsync(xs_dst, h->xs, sizeof(double) * h->B * (d.N + 1) * d.nx, hipMemcpyDeviceToDevice, h->stream));
    if (us_dst) UPR_HIP(hipMemcpyAsync(us_dst, h->us, sizeof(double) * h->B * d.N * d.nu, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int upr_batch_copy_policy_device(upr_batch* h, void* u_dst) {
    UPR_ENTER(h);
    if (!h->ev_u) return fail("upr_batch_copy_policy_device: no tick has run on this handle");
    if (!u_dst) return fail("upr_batch_copy_policy_device: null destination");
    UPR_HIP(hipMemcpyAsync(u_dst, h->ev_u, sizeof(double) * h->B * h->d.nu, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

void* upr_batch_stream(upr_batch* h) { return h ? (void*)h->stream : nullptr; }

int upr_set_device(int device) {
    if (need_device()) return 1;
    int n = 0;
    UPR_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("upr_set_device: no such device");
    UPR_HIP(hipSetDevice(device));
    return 0;
}
int upr_batch_device(const upr_batch* h) { return h ? h->device : -1; }
long long upr_batch_ws_doubles(const upr_batch* h) { return h ? (long long)h->d.ws_stride : -1; }

/* forget the previous solution without a host synchronisation (cold start for the next advance) */
int upr_batch_reset_async(upr_batch* h) {
    UPR_ENTER(h);
    h->has_prev = false;
    h->guess_set = false;
    return 0;
}


// ---- interface states: [robot x (3 nq), dynamic obstacle r v a (9 n_dyn)] ------------------------------------------------
// The kernels work on the robot state; the obstacle is uncontrolled (system_dynamics.h:29-39), so its part of every
// trajectory is the ballistic continuation of its observed state.  With n_dyn == 0 these wrappers are pass-throughs.
static void narrow_states(const upr_batch* h, const double* xf, size_t n, std::vector<double>& xr) {
    xr.resize(n * h->d.nx);
    for (size_t i = 0; i < n; ++i) std::memcpy(xr.data() + i * h->d.nx, xf + i * h->nxf, sizeof(double) * h->d.nx);
}
static void obstacle_after(const double* xo, double tau, double* out, int n_dyn = 1) {   // every obstacle of an instance: [n_dyn][9]
    for (int o = 0; o < n_dyn; ++o, xo += 9, out += 9)
        for (int i = 0; i < 3; ++i) { out[6 + i] = xo[6 + i]; out[3 + i] = xo[3 + i] + tau * xo[6 + i]; out[i] = xo[i] + tau * xo[3 + i] + 0.5 * tau * tau * xo[6 + i]; }
}

int upr_batch_set_observation(upr_batch* h, const double* t, int t_stride, const double* x) {
    UPR_ENTER(h);
    if (!h->P.n_dyn) return set_observation_core(h, t, t_stride, x);
    std::vector<double> xr;
    narrow_states(h, x, h->B, xr);
    const size_t nd9 = 9 * (size_t)h->P.n_dyn;
    for (int b = 0; b < h->B; ++b) std::memcpy(h->hdyn0.data() + (size_t)b * nd9, x + (size_t)b * h->nxf + h->d.nx, sizeof(double) * nd9);
    UPR_HIP(hipMemcpy(h->dyn0, h->hdyn0.data(), sizeof(double) * h->B * nd9, hipMemcpyHostToDevice));
    return set_observation_core(h, t, t_stride, xr.data());
}

int upr_batch_set_guess(upr_batch* h, const double* xs, const double* us) {
    UPR_ENTER(h);
    if (!h->P.n_dyn) return set_guess_core(h, xs, us);
    std::vector<double> xr;
    narrow_states(h, xs, (size_t)h->B * (h->d.N + 1), xr);
    return set_guess_core(h, xr.data(), us);
}

int upr_batch_get_solution(upr_batch* h, double* ts, double* xs, double* us) {
    UPR_ENTER(h);
    if (!h->P.n_dyn || !xs) return get_solution_core(h, ts, xs, us);
    const upr_dims& d = h->d;
    std::vector<double> xr((size_t)h->B * (d.N + 1) * d.nx);
    if (get_solution_core(h, ts, xr.data(), us)) return 1;
    for (int b = 0; b < h->B; ++b) for (int k = 0; k <= d.N; ++k) {
        double* o = xs + ((size_t)b * (d.N + 1) + k) * h->nxf;
        std::memcpy(o, xr.data() + ((size_t)b * (d.N + 1) + k) * d.nx, sizeof(double) * d.nx);
        obstacle_after(h->hdyn0.data() + (size_t)b * 9 * h->P.n_dyn, k * h->P.dt, o + d.nx, h->P.n_dyn);
    }
    return 0;
}

static void widen_eval(upr_batch* h, const double* t, int t_stride, const double* xr, double* x_out) {
    std::vector<double> tp(h->B);
    (void)hipMemcpy(tp.data(), h->tprev, sizeof(double) * h->B, hipMemcpyDeviceToHost);
    for (int b = 0; b < h->B; ++b) {
        double* o = x_out + (size_t)b * h->nxf;
        std::memcpy(o, xr + (size_t)b * h->d.nx, sizeof(double) * h->d.nx);
        double tau = t[(size_t)b * (t_stride ? 1 : 0)] - tp[b];
        obstacle_after(h->hdyn_prev.data() + (size_t)b * 9 * h->P.n_dyn, tau > 0.0 ? tau : 0.0, o + h->d.nx, h->P.n_dyn);
    }
}

int upr_batch_evaluate(upr_batch* h, const double* t, int t_stride, double* x_out, double* u_out) {
    UPR_ENTER(h);
    if (!h->P.n_dyn) return evaluate_core(h, t, t_stride, x_out, u_out);
    std::vector<double> xr((size_t)h->B * h->d.nx);
    if (evaluate_core(h, t, t_stride, xr.data(), u_out)) return 1;
    widen_eval(h, t, t_stride, xr.data(), x_out);
    return 0;
}

int upr_batch_evaluate_policy(upr_batch* h, const double* t, int t_stride, const double* x_obs, double* x_out, double* u_out) {
    UPR_ENTER(h);
    if (!h->P.n_dyn) return evaluate_policy_core(h, t, t_stride, x_obs, x_out, u_out);
    std::vector<double> xo, xr((size_t)h->B * h->d.nx);
    narrow_states(h, x_obs, h->B, xo);
    if (evaluate_policy_core(h, t, t_stride, xo.data(), xr.data(), u_out)) return 1;
    widen_eval(h, t, t_stride, xr.data(), x_out);
    return 0;
}

int upr_batch_tick(upr_batch* h, const double* t, int t_stride, const double* x, double* x_out, double* u_out, double* stats_out) {
    UPR_ENTER(h);
    if (!t || !x || !x_out || !u_out) return fail("upr_batch_tick: null argument");
    const upr_dims& d = h->d;
    const size_t B = (size_t)h->B, nx = (size_t)d.nx, nu = (size_t)d.nu, ndyn = 9 * (size_t)h->P.n_dyn;
    if (!h->ev_t && (dev_alloc(&h->ev_t, h->B) || dev_alloc(&h->ev_xo, B * nx) || dev_alloc(&h->ev_x, B * (nx + nu)))) return 1;
    h->ev_u = h->ev_x + B * nx;
    // pinned staging: [t B (+ pad)][x B nx][dyn B 9 n_dyn] in (the layout of the device block h->t0), [x B nx][u B nu][stats B NSTATS] out
    const size_t Bt = (B + 1) & ~(size_t)1;   // (the block of times is padded to an even count, as on the device)
    const size_t n_in = Bt + B * nx + B * ndyn, n_out = B * nx + B * nu + B * UPR_NSTATS;
    if (!h->pin) {
        UPR_HIP(hipHostMalloc((void**)&h->pin, sizeof(double) * (n_in + n_out), hipHostMallocDefault));
    }
    double* pt = h->pin; double* px = pt + Bt; double* pd = px + B * nx;
    double* ox = h->pin + n_in; double* ou = ox + B * nx; double* os = ou + B * nu;
    for (size_t b = 0; b < B; ++b) {
        pt[b] = t[b * (t_stride ? 1 : 0)];
        std::memcpy(px + b * nx, x + b * (size_t)h->nxf, sizeof(double) * nx);
        if (ndyn) { std::memcpy(pd + b * ndyn, x + b * (size_t)h->nxf + nx, sizeof(double) * ndyn); std::memcpy(h->hdyn0.data() + b * ndyn, pd + b * ndyn, sizeof(double) * ndyn); }
    }
    auto t0c = std::chrono::steady_clock::now();
    // the stream operations of one control period
    auto enqueue = [&]() -> int {
        // (round 5: one copy in and one out -- device and pinned host blocks have the same layout -- instead of three each: a copy
        //  node of the period's graph costs about as much as a small kernel launch)
        UPR_HIP(hipMemcpyAsync(h->t0, pt, sizeof(double) * n_in, hipMemcpyHostToDevice, h->stream));
        if (advance_impl(h)) return 1;
        // the policy at the observation: time t0, state x0 (both already on the device)
        if (h->fb) hipLaunchKernelGGL(evaluate_policy_kernel, dim3(h->B), dim3(64), 0, h->stream, h->dP, d, h->B, h->tprev, h->xs_prev, h->us_prev, h->fb, h->t0, h->x0, h->ev_x, h->ev_u);
        else hipLaunchKernelGGL(evaluate_kernel, dim3(h->B), dim3(64), 0, h->stream, h->dP, d, h->B, h->tprev, h->xs_prev, h->us_prev, h->t0, 1, h->ev_x, h->ev_u);
        UPR_HIP(hipGetLastError());
        UPR_HIP(hipMemcpyAsync(ox, h->ev_x, sizeof(double) * (B * nx + B * nu), hipMemcpyDeviceToHost, h->stream));
        if (stats_out) UPR_HIP(hipMemcpyAsync(os, h->stats, sizeof(double) * B * UPR_NSTATS, hipMemcpyDeviceToHost, h->stream));
        return 0;
    };
    if (h->tick_graph_on < 0) { const char* e = getenv("UPR_TICK_GRAPH"); h->tick_graph_on = (e && atoi(e) == 0) ? 0 : 1; }
    // what decides which operations a tick enqueues (advance_impl): warm start or not, a guess handed in, the SQP iteration count,
    // the dispatch order being valid, the feedback policy and who writes it, event timing, the statistics copy
    const int sqp_now = h->sqp_iters_next > 0 ? h->sqp_iters_next : h->P.sqp_iters;
    unsigned long long sig = 1469598103934665603ull;
    for (unsigned long long v : {(unsigned long long)h->has_prev, (unsigned long long)h->guess_set, (unsigned long long)sqp_now, (unsigned long long)(h->order_on && h->order_valid),
                                 (unsigned long long)(h->fb != nullptr), (unsigned long long)h->fb_fused, (unsigned long long)h->timing, (unsigned long long)(stats_out != nullptr),
                                 (unsigned long long)h->use_qp3, (unsigned long long)h->use_qp2, (unsigned long long)h->qp_nt, (unsigned long long)(uintptr_t)h->pin,
                                 (unsigned long long)(uintptr_t)h->prof /* (passed BY VALUE in upr_qp_args: a graph captured before upr_batch_qp_profile allocated it would keep nullptr) */}) { sig ^= v; sig *= 1099511628211ull; }
    const bool steady = h->tick_graph_on && !h->timing && h->has_prev && !h->guess_set && h->sqp_iters_next == 0 && (!h->order_on || h->order_valid);
    if (steady && h->tick_exec && sig == h->tick_sig) {
        UPR_HIP(hipGraphLaunch(h->tick_exec, h->stream));
        // (what advance_impl does on the host besides enqueueing)
        h->hdyn_prev = h->hdyn0;
        ++h->tick_replays;
    } else {
        if (h->tick_exec && sig != h->tick_sig) { (void)hipGraphExecDestroy(h->tick_exec); h->tick_exec = nullptr; h->tick_steady = 0; }
        h->tick_steady = (steady && sig == h->tick_sig) ? h->tick_steady + 1 : 0;
        h->tick_sig = sig;
        if (steady && h->tick_steady >= 2 && !h->tick_exec) {
            // third steady tick in a row: capture it, run the captured graph for this tick
            hipGraph_t g = nullptr;
            UPR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue();
            const hipError_t ec = hipStreamEndCapture(h->stream, &g);
            if (rc || ec != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); h->tick_graph_on = 0; if (rc) return 1; if (enqueue()) return 1; }
            else {
                const hipError_t ei = hipGraphInstantiate(&h->tick_exec, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (ei != hipSuccess) { h->tick_exec = nullptr; h->tick_graph_on = 0; if (enqueue()) return 1; }
                else UPR_HIP(hipGraphLaunch(h->tick_exec, h->stream));
            }
        } else if (enqueue()) return 1;
    }
    UPR_HIP(hipStreamSynchronize(h->stream));
    h->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0c).count();
    for (size_t b = 0; b < B; ++b) {
        double* o = x_out + b * (size_t)h->nxf;
        std::memcpy(o, ox + b * nx, sizeof(double) * nx);
        if (ndyn) std::memcpy(o + nx, pd + b * ndyn, sizeof(double) * ndyn);   // (evaluated at the observation's own time: the obstacle where it was observed)
    }
    std::memcpy(u_out, ou, sizeof(double) * B * nu);
    if (stats_out) std::memcpy(stats_out, os, sizeof(double) * B * UPR_NSTATS);
    return 0;
}

long long upr_batch_tick_graph_replays(upr_batch* h) { return h ? h->tick_replays : 0; }

int upr_batch_get_feedback(upr_batch* h, double* K) {
    UPR_ENTER(h);
    if (!h->P.n_dyn) return get_feedback_core(h, K);
    const upr_dims& d = h->d;
    std::vector<double> Kr((size_t)h->B * d.N * d.nu * d.nx);
    if (get_feedback_core(h, Kr.data())) return 1;
    const size_t rows = (size_t)h->B * d.N * d.nu;
    for (size_t r = 0; r < rows; ++r) {   // the obstacle is not fed back: zero columns
        std::memcpy(K + r * h->nxf, Kr.data() + r * d.nx, sizeof(double) * d.nx);
        for (int c = d.nx; c < h->nxf; ++c) K[r * h->nxf + c] = 0.0;
    }
    return 0;
}

int upr_batch_qp_step(upr_batch* h, double* dxs, double* dus) {
    UPR_ENTER(h);
    if (!h->P.n_dyn || !dxs) return qp_step_core(h, dxs, dus);
    const upr_dims& d = h->d;
    std::vector<double> xr((size_t)h->B * (d.N + 1) * d.nx);
    if (qp_step_core(h, xr.data(), dus)) return 1;
    const size_t n = (size_t)h->B * (d.N + 1);
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(dxs + i * h->nxf, xr.data() + i * d.nx, sizeof(double) * d.nx);
        for (int c = d.nx; c < h->nxf; ++c) dxs[i * h->nxf + c] = 0.0;
    }
    return 0;
}

int upr_batch_set_projectile_flag(upr_batch* h, const double* sflag) {
    UPR_ENTER(h);
    if (!h->pflag) return fail("upr_batch_set_projectile_flag: the problem has no dynamic obstacle");
    UPR_HIP(hipMemcpy(h->pflag, sflag, sizeof(double) * h->B, hipMemcpyHostToDevice));
    return 0;
}
}  // extern "C"

#ifdef UPR_LS_PROF
// instrumented build only: cycles from the start of the line-search kernel to each of its stamps, summed over workgroups, [15] = workgroups
extern "C" int upr_debug_ls_prof(double* out, int reset) {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(upr_ls_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = (double)h[i];
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(upr_ls_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifdef UPR_LIN_PROF
// instrumented build only: cycles per phase of the linearisation kernel summed over workgroups, [7] = workgroups counted
extern "C" int upr_debug_lin_prof(double* out, int reset) {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(upr_lin_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(upr_lin_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
