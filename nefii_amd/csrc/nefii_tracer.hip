// nefii_tracer.hip - SDF ray tracer for gfx950: RayTracing.forward of the reference
// (code/model/ray_tracing.py:29-337) restated as a per-ray state machine driven in ROUNDS.
//
// Every round is two launches on one stream, with no host synchronisation:
//   advance : one thread per ray.  Consumes the SDF values its ray asked for in the previous round,
//             moves the ray's state machine (sphere tracing both ends + back-off line search ->
//             100-sample bracket search -> bisection -> min-SDF search) and appends the ray's next
//             SDF queries to a compacted work list (wave ballot + prefix popcount, one atomic per block).
//   eval    : the fused SDF MLP (mlp_tile.h) over dense 32-query tiles of that list - rays in different
//             phases share tiles, so the matrix cores only ever see live queries.
// Ray state is a few floats per ray in global memory (L2 resident); all heavy traffic is the MLP.
// Arithmetic order follows the reference exactly (separate multiply and add for o + t*d etc.); the
// only semantic difference is that bisection stops per ray instead of when the whole batch converged.
#include <cstdlib>
#include <vector>

#include "mlp_tile.h"

using namespace nefii;

#define HIP_CHECK_LAUNCH()                       \
    do {                                         \
        hipError_t _e = hipGetLastError();       \
        if (_e != hipSuccess) return (int)_e;    \
    } while (0)

namespace {

// PH_SAMPLER_C / PH_MINSDF_C: the ray's n_steps samples are being evaluated by the single-pass (coarse) evaluator; the
// exact phases PH_SAMPLER / PH_MINSDF follow once the samples that decide have been re-evaluated in split precision
// PH_SAMPLER_X: the bracket search's FIRST `chunk` samples are being evaluated in split precision (through the refine
// list) before anything else: sphere tracing stops right in front of the surface, so the first negative sample is one of the
// first few for most rays that have one (median index 1-3 of 100 on the bench scenes) - and the reference's decision then
// depends on no later sample.  A ray without a negative sample among them goes on to PH_SAMPLER_C as before.
enum Phase : int { PH_DONE = 0, PH_TRACE = 1, PH_SAMPLER = 2, PH_BISECT = 3, PH_MINSDF = 4, PH_SAMPLER_C = 5, PH_MINSDF_C = 6,
                   PH_SAMPLER_X = 7 };
constexpr int PH_POST = 100;      // local to advance_kernel: the stage behind tracing / sampler / bisection
constexpr int NCNT = NEFII_TRACE_COUNTERS;
constexpr int NEAR_PROBES = 6;     // skipped samples per staged search that the audit evaluates because their bound cleared the limit by < 2 tau
enum Kind : int { Q_START = 0, Q_END = 1, Q_MID = 2 };

struct RayState {            // SoA views into the workspace
    float *t_s, *t_e, *cur_s, *cur_e, *nxt_s, *nxt_e, *t_min, *t_max, *res_s, *res_e, *lo, *hi, *mid;
    int *flags;              // phase | live/pending bits | counters (packed, see below)
    float *big;              // [n][n_steps] dense query results
    unsigned *singles;       // [2n]  (ray << 2 | kind)
    unsigned *dense;         // [n]   (ray << 1 | which)  which: 0 sampler, 1 min-sdf
    unsigned *tri;           // [n]   rays in bisection: 7 speculative queries each (3 levels of the bisection tree)
    unsigned *cdense;        // [4n]  (window << 29 | ray << 1 | which): quarter rows (CW samples of a ray's n_steps) for the coarse evaluator
    unsigned *refine;        // [n * cap]  (ray << 7 | sample): coarse samples to re-evaluate in split precision
    unsigned *csingles;      // [2n]  (ray << 2 | kind): sphere-tracing queries for the coarse evaluator (tiered sphere tracing)
    int *block_live;         // [ceil(n / 256)]  advance_kernel: does this block still hold a ray that is not done?
    unsigned *crefine;       // [n * CREF_CAP]  (ray << 7 | sample): single samples for the COARSE evaluator (staged min-SDF search)
    unsigned char *ord;      // [groups][n_steps]  staged min-SDF search: ord[k] = index of the k-th smallest of the group's draws
};
// staged searches: most second-stage samples of one ray (a ray with more takes its whole row, as without the staging)
constexpr int CREF_CAP = 76;       // (n_steps 100: the 75 depths outside the first stage + the probe - no search falls back)

// flags layout
constexpr int F_PHASE = 0x7;          // bits 0-2
constexpr int F_LIVE_S = 1 << 3, F_LIVE_E = 1 << 4, F_PEND_S = 1 << 5, F_PEND_E = 1 << 6;
constexpr int F_STEPPED = 1 << 7;     // results belong to a step / back-off (not the initial evaluation)
constexpr int F_SPH = 1 << 8, F_SAMP = 1 << 9, F_HIT = 1 << 10;
constexpr int F_IT_SHIFT = 12, F_IT_MASK = 0xFF;     // sphere-tracing iteration / bisection step
constexpr int F_K_SHIFT = 20, F_K_MASK = 0xF;        // back-off count
// PH_SAMPLER_C only: 0 = the whole row is with the coarse evaluator; w = 1..4: windowed search, the first w quarter rows are
// (PH_SAMPLER_C also: 5 / 6 = first / second stage of the staged bracket search (minsdf_lipschitz) is with the coarse evaluator)
// PH_MINSDF_C: 0 = the row's coarse values are in, 1 = second stage of the two-stage refinement, 2 / 3 = first / second stage
// of the staged search (minsdf_lipschitz) is with the coarse evaluator
constexpr int F_WIN_SHIFT = 24, F_WIN_MASK = 0x7;
constexpr int CWIN_STAGE1 = 4;        // cdense window code: the first stage's depths of a staged min-SDF search
// Tiered sphere tracing (nefii_tracer_params.trace_tier), PH_TRACE only.  F_CRS_x: the pending result of that end comes from
// the single-pass evaluator.  F_AUD_x: that end's query is being REPEATED in split precision this round and res_x still
// holds the coarse value (the split evaluator compares the two: the online audit of coarse_tau).
constexpr int F_CRS_S = 1 << 27, F_CRS_E = 1 << 28, F_AUD_S = 1 << 29, F_AUD_E = 1 << 30;
// samples per quarter row (window) of a coarse row
__host__ __device__ __forceinline__ int coarse_window(int n_steps) { return (n_steps + 3) >> 2; }

__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

struct Params {
    nefii_tracer_params p;
    int64_t n;
    const float *o, *d;
    const uint8_t *obj;
    const float *lin, *steps;
    float *out_pts, *out_dist;
    uint8_t *out_hit;
    int *counters;           // [rounds][4]
    int levels, tri_nodes;   // speculative bisection: levels per round, nodes = 2^levels - 1
    float tau;               // coarse pass: error bound of a coarse sample (0: coarse pass off)
    int cap;                 //              most samples of one ray refined individually
    int chunk;               //              leading samples of a bracket search evaluated exactly first (0: off)
    float chunk_gate;        //              ... for rays whose front SDF is below chunk_gate x the chunk's reach
    int window;              //              bracket searches inside the object mask take their coarse samples a quarter row at a time
    float tier_band;         // tiered sphere tracing: a coarse value v16 decides (v > thr, sign) when |v16| > tier_band; 0: off
    float tier_gate;         //              a step / back-off query goes to the coarse evaluator when the step that led to it is > tier_gate
    float lip;               // staged min-SDF search: Lipschitz bound of the SDF along a ray (0: off)
    int stage_bracket;       //              ... and the bracket search of eval-mode traces / rays outside the mask staged too
    int miss_argmin;         // eval-mode bracket search: the argmin fallback of rays WITHOUT a negative sample is computed (1; 0: nobody reads it)
    RayState s;
};

// first stage of the staged min-SDF search: sorted position of its j-th depth, j < stage1_count (both ends included)
__host__ __device__ __forceinline__ int stage1_count(int ns) { return coarse_window(ns); }
__host__ __device__ __forceinline__ int stage1_pos(int ns, int j) { return (j * (ns - 1)) / (stage1_count(ns) - 1); }
// the staged bracket search takes the rays whose quarter-row windows are long searches: eval-mode traces (the secondary rays
// of the MC renderer, full-frame renders - most of their searches find their crossing late or not at all) and rays outside
// the object mask (whole rows, argmin); training-mode rays inside the mask stop in their first window and keep the windows
__device__ __forceinline__ bool staged_bracket(const Params &P, int64_t r) {
    return P.lip > 0.f && P.stage_bracket && (!P.p.training || P.obj[r] == 0);
}
__device__ __forceinline__ int minsdf_row(const Params &P, int64_t r) {
    const int g = P.p.minsdf_group;
    return g > 0 ? (int)(r / g) : 0;
}

// uniform draw i of ray r's min-SDF search: one row for the whole call, or one row per minsdf_group consecutive rays
// (several batches traced as one call keep their own draws)
__device__ __forceinline__ float minsdf_step(const Params &P, int64_t r, int i) {
    const int g = P.p.minsdf_group;
    return P.steps[(g > 0 ? (r / g) * (int64_t)P.p.n_steps : 0) + i];
}

// ---- work-list append: block-aggregated ------------------------------------------------------
// each thread contributes up to 2 single queries, one dense ray (split precision or coarse), one bisecting ray and
// n_ref coarse samples to refine (bit set `cmask`).
// nc / cwin: quarter rows cwin .. cwin + nc - 1 of the ray's samples for the coarse evaluator (4 from 0: the whole row).
// qcs / qce: the ray's start / end query for the COARSE evaluator (tiered sphere tracing); n_rep: split-precision singles of
// this ray that repeat a coarse one
__device__ __forceinline__ void append_queries(const Params &P, int round, bool qs, bool qe, bool qt, bool qd, int nc, int cwin,
                                               unsigned ray, unsigned dense_which, int consumed, int n_alg, int n_ref,
                                               const unsigned (&cmask)[4], bool qcs = false, bool qce = false, int n_rep = 0,
                                               bool ref_coarse = false) {
    // ref_coarse: this ray's n_ref samples go to the COARSE evaluator's list (crefine: staged min-SDF search)
    __shared__ int wtot[10][4];
    __shared__ int base[7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bs = __ballot(qs), be = __ballot(qe), bt = __ballot(qt), bd = __ballot(qd);
    const unsigned long long bcs = __ballot(qcs), bce = __ballot(qce);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cons = consumed, alg = n_alg, rep = n_rep;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cons += __shfl_xor(cons, o), alg += __shfl_xor(alg, o), rep += __shfl_xor(rep, o);
    int incl = ref_coarse ? 0 : n_ref;      // inclusive prefix sums of the two refine counts over the wave
    int incl2 = ref_coarse ? n_ref : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o), t2 = __shfl_up(incl2, o);
        if (lane >= o) incl += t, incl2 += t2;
    }
    const int wref = __shfl(incl, 63), wref2 = __shfl(incl2, 63);
    int cincl = nc;                 // ... and of the coarse window counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(cincl, o);
        if (lane >= o) cincl += t;
    }
    const int wcoarse = __shfl(cincl, 63);
    if (lane == 0) {
        wtot[0][wave] = __popcll(bs) + __popcll(be);
        wtot[1][wave] = __popcll(bd);
        wtot[2][wave] = __popcll(bt);
        wtot[3][wave] = cons;
        wtot[4][wave] = wref;
        wtot[5][wave] = wcoarse;
        wtot[6][wave] = alg;
        wtot[7][wave] = __popcll(bcs) + __popcll(bce);
        wtot[8][wave] = rep;
        wtot[9][wave] = wref2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t[10];
        for (int i = 0; i < 10; ++i) t[i] = wtot[i][0] + wtot[i][1] + wtot[i][2] + wtot[i][3];
        int *cnt = P.counters + round * NCNT;
        for (int i = 0; i < 3; ++i) base[i] = t[i] ? atomicAdd(&cnt[i], t[i]) : 0;
        if (t[3]) atomicAdd(&cnt[3], t[3]);
        base[3] = t[4] ? atomicAdd(&cnt[4], t[4]) : 0;
        base[4] = t[5] ? atomicAdd(&cnt[5], t[5]) : 0;
        if (t[6]) atomicAdd(&cnt[6], t[6]);
        if (t[2]) atomicAdd(&cnt[7], t[2] * P.tri_nodes);      // speculative bisection evaluations executed
        base[5] = t[7] ? atomicAdd(&cnt[9], t[7]) : 0;
        if (t[8]) atomicAdd(&cnt[10], t[8]);
        base[6] = t[9] ? atomicAdd(&cnt[11], t[9]) : 0;
    }
    __syncthreads();
    int off_s = base[0], off_d = base[1], off_t = base[2], off_r = base[3], off_c = base[4], off_cs = base[5], off_r2 = base[6];
    for (int w = 0; w < wave; ++w) {
        off_r2 += wtot[9][w];
        off_s += wtot[0][w];
        off_d += wtot[1][w];
        off_t += wtot[2][w];
        off_r += wtot[4][w];
        off_c += wtot[5][w];
        off_cs += wtot[7][w];
    }
    if (qs) P.s.singles[off_s + __popcll(bs & lt)] = (ray << 2) | Q_START;
    if (qe) P.s.singles[off_s + __popcll(bs) + __popcll(be & lt)] = (ray << 2) | Q_END;
    if (qcs) P.s.csingles[off_cs + __popcll(bcs & lt)] = (ray << 2) | Q_START;
    if (qce) P.s.csingles[off_cs + __popcll(bcs) + __popcll(bce & lt)] = (ray << 2) | Q_END;
    if (qd) P.s.dense[off_d + __popcll(bd & lt)] = (ray << 1) | dense_which;
    if (qt) P.s.tri[off_t + __popcll(bt & lt)] = ray;
    for (int k = 0; k < nc; ++k)
        P.s.cdense[off_c + cincl - nc + k] = ((unsigned)(cwin + k) << 29) | (ray << 1) | dense_which;
    if (n_ref > 0) {
        size_t o = ref_coarse ? (size_t)off_r2 + incl2 - n_ref : (size_t)off_r + incl - n_ref;
        unsigned *list = ref_coarse ? P.s.crefine : P.s.refine;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned mbits = cmask[w];
            while (mbits) {
                const int bit = __ffs(mbits) - 1;
                mbits &= mbits - 1;
                list[o++] = (ray << 7) | (unsigned)(32 * w + bit);
            }
        }
    }
    __syncthreads();
}

// depth of heap node j (root 0, children 2j+1 / 2j+2) of the speculative bisection tree over [lo, hi]: replay the
// recurrence mid = (lo + hi) / 2 (ray_tracing.py:262,275) along the node's path - the speculated points ARE the
// points sequential bisection would visit.  Path bits, MSB first after the leading 1 of (j+1): 1 = "f(mid) > 0"
// (lo = mid), 0 = hi = mid.
__device__ __forceinline__ float tri_depth(float lo, float hi, int j) {
    const unsigned k = (unsigned)j + 1u;
    const int level = 31 - __clz(k);
    float mid = fmul(fadd(lo, hi), 0.5f);
    for (int b = level - 1; b >= 0; --b) {
        if ((k >> b) & 1u) lo = mid; else hi = mid;
        mid = fmul(fadd(lo, hi), 0.5f);
    }
    return mid;
}

__device__ __forceinline__ void finish(const Params &P, int64_t r, float dist, bool hit) {
    const float ox = P.o[r * 3], oy = P.o[r * 3 + 1], oz = P.o[r * 3 + 2];
    const float dx = P.d[r * 3], dy = P.d[r * 3 + 1], dz = P.d[r * 3 + 2];
    P.out_dist[r] = dist;
    P.out_hit[r] = hit ? 1 : 0;
    P.out_pts[r * 3] = fadd(ox, fmul(dist, dx));
    P.out_pts[r * 3 + 1] = fadd(oy, fmul(dist, dy));
    P.out_pts[r * 3 + 2] = fadd(oz, fmul(dist, dz));
}

// staged min-SDF search: sorted order of every row of draws (rank by value, ties by index), once per call
__global__ __launch_bounds__(128) void minsdf_order_kernel(Params P) {
    const int ns = P.p.n_steps, i = threadIdx.x;
    if (i >= ns) return;
    const float *st = P.steps + (size_t)blockIdx.x * ns;
    const float si = st[i];
    int rank = 0;
    for (int j = 0; j < ns; ++j) {
        const float sj = st[j];
        rank += (sj < si || (sj == si && j < i)) ? 1 : 0;
    }
    P.s.ord[(size_t)blockIdx.x * ns + rank] = (unsigned char)i;
}

// ---- the per-ray state machine ---------------------------------------------------------------
__global__ __launch_bounds__(256) void advance_kernel(Params P, int round) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < P.n;
    bool qs = false, qe = false, qt = false, qd = false;
    int nc = 0, cwin = 0;         // quarter rows of this ray for the coarse evaluator
    unsigned dense_which = 0;
    int consumed = 0;          // bisection evaluations actually used this round (of the 7 speculated per ray)
    int n_alg = 0;             // dense searches entered this round (the reference evaluates n_steps samples for each)
    int n_ref = 0;             // coarse samples of this ray to re-evaluate in split precision (bits of cmask)
    unsigned cmask[4] = {0u, 0u, 0u, 0u};
    bool cs = false, ce = false;       // tiered sphere tracing: qs / qe go to the coarse evaluator
    int n_rep = 0;                     //                         split-precision singles that repeat a coarse one
    bool ref_coarse = false;           // staged min-SDF search: the n_ref samples of cmask go to the coarse evaluator
    const bool tier = P.tier_band > 0.f;
    const bool coarse = P.tau > 0.f;
    const nefii_tracer_params &tp = P.p;
    const float thr = tp.sdf_threshold;
    // a block whose rays are all done has nothing to advance and nothing to append (late rounds - the bisection's - keep a
    // few percent of a big batch's rays: config 3 spent 3.4 ms per step in this kernel)
    if (round > 0 && P.s.block_live[blockIdx.x] == 0) return;
    int fl = 0;
    if (valid) fl = P.s.flags[r];
    int ph = fl & F_PHASE;

    if (valid && round == 0) {
        // bounding-sphere intersection (rend_util.py:200-221) and state initialisation (:107-134)
        const float ox = P.o[r * 3], oy = P.o[r * 3 + 1], oz = P.o[r * 3 + 2];
        const float dx = P.d[r * 3], dy = P.d[r * 3 + 1], dz = P.d[r * 3 + 2];
        const float b = fadd(fadd(fmul(dx, ox), fmul(dy, oy)), fmul(dz, oz));
        const float nrm = sqrtf(fadd(fadd(fmul(ox, ox), fmul(oy, oy)), fmul(oz, oz)));
        const float rad = tp.object_bounding_sphere;
        const float under = fsub(fmul(b, b), fsub(fmul(nrm, nrm), fmul(rad, rad)));
        const bool sph = under > 0.f;
        float t0 = 0.f, t1 = 0.f;
        if (sph) {
            const float sq = sqrtf(under);
            t0 = fmaxf(fsub(-sq, b), 0.01f);
            t1 = fmaxf(fsub(sq, b), 0.01f);
        }
        P.s.t_s[r] = t0;
        P.s.t_e[r] = t1;
        P.s.t_min[r] = t0;
        P.s.t_max[r] = t1;
        P.s.nxt_s[r] = 0.f;
        P.s.nxt_e[r] = 0.f;
        P.s.cur_s[r] = 0.f;
        P.s.cur_e[r] = 0.f;
        fl = PH_TRACE;
        if (sph) {
            fl |= F_SPH | F_LIVE_S | F_LIVE_E | F_PEND_S | F_PEND_E;
            qs = qe = true;
            // tiered sphere tracing: the first evaluations lie on the bounding sphere, far from the surface - unless the
            // ray starts inside it (secondary rays: t0 clamped to 0.01 off the surface they leave)
            cs = tier && t0 > 0.01f;
            ce = tier && t1 > 0.01f;
            fl |= (cs ? F_CRS_S : 0) | (ce ? F_CRS_E : 0);
        }
        ph = PH_TRACE;
        if (sph) {
            P.s.flags[r] = fl;
            ph = -1;     // wait for the first evaluation
        }
    }

    if (valid && ph == PH_TRACE) {
        float t_s = P.s.t_s[r], t_e = P.s.t_e[r];
        float cur_s = P.s.cur_s[r], cur_e = P.s.cur_e[r];
        float nxt_s = P.s.nxt_s[r], nxt_e = P.s.nxt_e[r];
        bool live_s = fl & F_LIVE_S, live_e = fl & F_LIVE_E;
        int it = (fl >> F_IT_SHIFT) & F_IT_MASK, k = (fl >> F_K_SHIFT) & F_K_MASK;
        if (fl & F_PEND_S) nxt_s = P.s.res_s[r];
        if (fl & F_PEND_E) nxt_e = P.s.res_e[r];
        // Tiered sphere tracing (nefii_tracer_params.trace_tier).  What the recurrence decides with an SDF value v is
        // `v <= thr` (the front has arrived) and `v < 0` (back off); a value v16 of the single-pass evaluator, |v16 - v| <
        // tau, decides both the same way when |v16| > tier_band (>= tau + thr) - it is then taken AS the value: the front
        // advances by v16 instead of v (not bit-identical to the split-precision trace: DESIGN, "tiered sphere tracing").
        // Inside the band the same query is repeated in split precision before anything moves.
        fl &= ~(F_AUD_S | F_AUD_E);
        const bool rep_s = (fl & F_PEND_S) && (fl & F_CRS_S) && !(fabsf(nxt_s) > P.tier_band);
        const bool rep_e = (fl & F_PEND_E) && (fl & F_CRS_E) && !(fabsf(nxt_e) > P.tier_band);
        bool wait = false;
        if (rep_s || rep_e) {
            if (rep_s) qs = true, fl = (fl & ~F_CRS_S) | F_AUD_S, ++n_rep;
            if (rep_e) qe = true, fl = (fl & ~F_CRS_E) | F_AUD_E, ++n_rep;
            P.s.flags[r] = fl;         // everything else as it is: the other end's result stays in res_x and is read again
            ph = -1;
        } else {
        fl &= ~(F_CRS_S | F_CRS_E);
        if (fl & F_STEPPED) {
            // back-off line search for ends that crossed the surface (ray_tracing.py:170-188)
            const bool bad_s = nxt_s < 0.f, bad_e = nxt_e < 0.f;
            if ((bad_s || bad_e) && k < tp.line_step_iters) {
                const float back = (1.f - tp.line_search_step) / (float)(1 << k);
                fl &= ~(F_PEND_S | F_PEND_E);
                if (bad_s) {
                    t_s = fsub(t_s, fmul(back, cur_s));
                    qs = true;
                    cs = tier && back * cur_s > P.tier_gate;
                    fl |= F_PEND_S | (cs ? F_CRS_S : 0);
                }
                if (bad_e) {
                    t_e = fadd(t_e, fmul(back, cur_e));
                    qe = true;
                    ce = tier && back * cur_e > P.tier_gate;
                    fl |= F_PEND_E | (ce ? F_CRS_E : 0);
                }
                ++k;
                wait = true;
            } else {
                live_s = live_s && (t_s < t_e);
                live_e = live_e && (t_s < t_e);
            }
        }
        if (!wait) {
            // loop top (ray_tracing.py:136-157)
            cur_s = live_s ? nxt_s : 0.f;
            if (cur_s <= thr) cur_s = 0.f;
            cur_e = live_e ? nxt_e : 0.f;
            if (cur_e <= thr) cur_e = 0.f;
            live_s = live_s && (cur_s > thr);
            live_e = live_e && (cur_e > thr);
            if (it == tp.sphere_tracing_iters || !(live_s || live_e)) {
                // tracing finished for this ray (ray_tracing.py:43-64)
                const bool hit = t_s < t_e;
                // (the K field is free from here on: it keeps the iterations this ray's sphere tracing took - read by
                // tools/tier_parity.py from the workspace, by nothing else)
                fl = (fl & F_SPH) | (hit ? F_HIT : 0) | (live_s ? F_SAMP : 0) | ((it & F_K_MASK) << F_K_SHIFT);
                P.s.t_s[r] = t_s;
                P.s.t_e[r] = t_e;
                if (live_s) {
                    // what is left between the two fronts is often a short stretch hugging the surface (a grazing ray):
                    // the SDF is ~1-Lipschitz, so |sdf| <= (sdf_s + sdf_e + length) / 2 on it - when that is within the
                    // coarse pass's error bound nearly every sample would have to be refined, and the samples go to the
                    // split evaluator directly (a performance choice only: either way decides from exact values)
                    const bool go_coarse = coarse && 0.5f * (cur_s + cur_e + (t_e - t_s)) > 3.f * P.tau;
                    // inside the object mask (outside it the argmin over ALL samples is the result) the first `chunk`
                    // samples go ahead in split precision: PH_SAMPLER_X
                    // ... when a negative sample among them is plausible: the SDF at the front (cur_s, ~ the distance to
                    // the surface) is within reach of the chunk's last sample (a heuristic about WHICH path is cheaper -
                    // either way decides from exact values)
                    if (go_coarse && P.chunk > 0 && P.obj[r] != 0 &&
                        cur_s <= P.chunk_gate * (float)(P.chunk - 1) * (t_e - t_s) / (float)(tp.n_steps - 1)) {
                        fl |= PH_SAMPLER_X;
                        n_ref = P.chunk;
                        cmask[0] = (1u << P.chunk) - 1u;
                    } else {
                        fl |= go_coarse ? PH_SAMPLER_C : PH_SAMPLER;
                        qd = !go_coarse;
                        // inside the object mask the search ends at the first negative sample: the quarter rows go out one
                        // at a time (P.window; outside the mask the argmin over the whole row is the result)
                        if (go_coarse && staged_bracket(P, r)) {
                            // staged search (minsdf_lipschitz): a quarter row's worth of samples spread over the row first
                            float *v = P.s.big + (size_t)r * tp.n_steps;
                            for (int i = 0; i < tp.n_steps; ++i) v[i] = __builtin_inff();
                            nc = 1;
                            cwin = CWIN_STAGE1;
                            fl |= 5 << F_WIN_SHIFT;
                        } else if (go_coarse && (P.window & 1) && P.obj[r] != 0) {
                            nc = 1;
                            fl |= 1 << F_WIN_SHIFT;
                        } else if (go_coarse) {
                            nc = 4;
                        }
                    }
                    n_alg = 1;
                    dense_which = 0;
                    P.s.flags[r] = fl;
                    ph = -1;
                } else {
                    P.s.mid[r] = t_s;      // dist so far
                    ph = PH_POST;          // falls through to the post-sampler stage below
                }
            } else {
                ++it;
                t_s = fadd(t_s, cur_s);
                t_e = fsub(t_e, cur_e);
                nxt_s = 0.f;
                nxt_e = 0.f;
                k = 0;
                fl &= ~(F_PEND_S | F_PEND_E);
                if (live_s) {
                    qs = true;
                    cs = tier && cur_s > P.tier_gate;     // the step just taken ~ the distance the front was from the surface
                    fl |= F_PEND_S | (cs ? F_CRS_S : 0);
                }
                if (live_e) {
                    qe = true;
                    ce = tier && cur_e > P.tier_gate;
                    fl |= F_PEND_E | (ce ? F_CRS_E : 0);
                }
                fl |= F_STEPPED;
                wait = true;
            }
        }
        if (wait) {
            fl &= ~(F_LIVE_S | F_LIVE_E | (F_IT_MASK << F_IT_SHIFT) | (F_K_MASK << F_K_SHIFT));
            fl |= (live_s ? F_LIVE_S : 0) | (live_e ? F_LIVE_E : 0) | (it << F_IT_SHIFT) | (k << F_K_SHIFT);
            P.s.t_s[r] = t_s;
            P.s.t_e[r] = t_e;
            P.s.cur_s[r] = cur_s;
            P.s.cur_e[r] = cur_e;
            P.s.nxt_s[r] = nxt_s;
            P.s.nxt_e[r] = nxt_e;
            P.s.flags[r] = fl;
            ph = -1;
        }
        }       // (not repeating a coarse query)
    }

    if (valid && ph == PH_SAMPLER_X) {
        // the first `chunk` samples hold EXACT values.  A negative one among them at index >= 1 (index 0 would pair with
        // the LAST sample, ray_tracing.py:245-246) settles the search: `ind` is the first negative sample, the bracket is
        // (ind - 1, ind), the ray has a hit inside the object mask, so neither the argmin nor any later sample is read -
        // the exact stage below runs on these values with the rest of the row out of the way.  Otherwise: all n_steps
        // samples through the coarse evaluator, as if this stage had not been.
        const int ns = tp.n_steps;
        float *v = P.s.big + (size_t)r * ns;
        int ind = -1;
        for (int i = 0; i < P.chunk; ++i)
            if (v[i] < 0.f && ind < 0) ind = i;
        if (ind >= 1) {
            for (int i = P.chunk; i < ns; ++i) v[i] = 3.0e38f;
            ph = PH_SAMPLER;
        } else {
            fl = (fl & ~F_PHASE) | PH_SAMPLER_C;
            if (staged_bracket(P, r)) {
                // (the leading samples keep their exact values: evaluated samples like any other)
                for (int i = P.chunk; i < ns; ++i) v[i] = __builtin_inff();
                nc = 1;
                cwin = CWIN_STAGE1;
                fl |= 5 << F_WIN_SHIFT;
            } else if (P.window & 1) {         // (PH_SAMPLER_X rays lie inside the object mask)
                nc = 1;
                fl |= 1 << F_WIN_SHIFT;
            } else {
                nc = 4;
            }
            dense_which = 0;
            P.s.flags[r] = fl;
            ph = -1;
        }
    }

    if (valid && ph == PH_SAMPLER_C && ((fl >> F_WIN_SHIFT) & F_WIN_MASK) >= 5) {
        // Staged bracket search (nefii_tracer_params.minsdf_lipschitz).  What the search decides: the FIRST negative sample
        // and - unless the ray lies inside the object mask and surely has one - the argmin.  Lower bound of sample s between
        // evaluated neighbours a < s < b as in the staged min-SDF search below: v_s >= max(c_a - L dt_a, c_b - L dt_b) - tau.
        //   stage 1 done (5): with a surely negative first-stage sample j1 on a ray inside the mask only the samples in front
        //     of j1 matter, and only their sign: s is skipped for good when its bound is > 0.  Otherwise s must also not be
        //     the argmin: skipped when its bound exceeds max(0, best + tau).  The others go to the coarse evaluator one by
        //     one (with one skipped sample per search as the audit's probe);
        //   stage 2 done (6): audit, then the row - skipped samples at +inf: positive, never a minimum - is decided as a whole.
        const int stage = (fl >> F_WIN_SHIFT) & F_WIN_MASK;
        const int ns = tp.n_steps, n1 = stage1_count(ns);
        float *v = P.s.big + (size_t)r * ns;
        const float Llen = fmul(P.lip, fsub(P.s.t_e[r], P.s.t_s[r]));
        const bool obj = P.obj[r] != 0;
        float best = __builtin_inff();
        int j1 = ns;
        for (int j = 0; j < n1; ++j) {
            const int i = stage1_pos(ns, j);
            best = fminf(best, v[i]);
            if (v[i] < -P.tau && j1 == ns) j1 = i;
        }
        const bool front_only = obj && j1 < ns && !(v[0] < P.tau);      // (sample 0 possibly negative: the row is decided as a whole)
        const bool sign_only = front_only || !P.miss_argmin;
        const float lim = sign_only ? 0.f : fmaxf(0.f, fadd(best, P.tau));
        const int last = front_only ? j1 : ns - 1;
        int ja = 0, k = 0, n_near = 0;
        float worst = 0.f;
        int probe = -1;
        unsigned probe_h = ~0u;
        for (int kk = 1; kk < ns - 1; ++kk) {
            if (kk == stage1_pos(ns, ja + 1)) {
                ++ja;
                continue;
            }
            const int ia = stage1_pos(ns, ja), ib = stage1_pos(ns, ja + 1);
            const float lb = fsub(fmaxf(fsub(v[ia], fmul(Llen, fsub(P.lin[kk], P.lin[ia]))),
                                        fsub(v[ib], fmul(Llen, fsub(P.lin[ib], P.lin[kk])))), P.tau);
            if (stage == 5) {
                if (kk >= last || v[kk] < __builtin_inff()) continue;      // not needed / one of the leading exact samples
                if (!(fsub(lb, 1e-6f) > lim)) {
                    cmask[kk >> 5] |= 1u << (kk & 31);
                    ++k;
                } else if (n_near < NEAR_PROBES && !(fsub(lb, 1e-6f) > fadd(lim, fmul(2.f, P.tau)))) {
                    // skipped, but its bound clears the limit by less than 2 tau: the skipped samples closest to mattering are
                    // evaluated after all - as PROBES of the audit (stage 6 holds every evaluated sample against its bound); a
                    // probe's value decides nothing the bound had not decided (ABI 15, counter 13)
                    cmask[kk >> 5] |= 1u << (kk & 31);
                    ++k, ++n_near;
                } else {
                    const unsigned h = ((unsigned)r * 2654435761u) ^ ((unsigned)(kk + 1) * 0x9E3779B1u);
                    const unsigned hh = (h ^ (h >> 15)) * 0x85EBCA6Bu;
                    if (hh < probe_h) probe_h = hh, probe = kk;
                }
            } else if (v[kk] < __builtin_inff()) {
                worst = fmaxf(worst, fsub(fsub(lb, P.tau), v[kk]));
            }
        }
        fl &= ~(F_WIN_MASK << F_WIN_SHIFT);
        if (stage == 5 && probe >= 0) {
            cmask[probe >> 5] |= 1u << (probe & 31);
            ++k, ++n_near;
        }
        if (stage == 5 && n_near > 0) atomicAdd(P.counters + round * NCNT + 13, n_near);
        if (stage == 5 && k > 0) {
            if (k <= CREF_CAP) {
                n_ref = k;
                ref_coarse = true;
                fl |= 6 << F_WIN_SHIFT;
            } else {        // too many to list: the whole row
                cmask[0] = cmask[1] = cmask[2] = cmask[3] = 0u;
                nc = 4;
            }
            dense_which = 0;
            P.s.flags[r] = fl;
            ph = -1;
        } else {
            // (the leading exact samples are audited with the rest: an exact value obeys the bound a fortiori)
            if (worst > 0.f) atomicMax(P.counters + round * NCNT + 12, __float_as_int(worst));
            cmask[0] = cmask[1] = cmask[2] = cmask[3] = 0u;      // the row's values are in: decided below, this round
        }
    }

    if (valid && ph == PH_SAMPLER_C && ((fl >> F_WIN_SHIFT) & F_WIN_MASK) != 0) {
        // Windowed search (rays inside the object mask): the first wn quarter rows hold coarse values.  A sample that is
        // SURELY negative (< -tau) among them ends the search exactly as the whole row would: the decision below reads the
        // samples up to it only (no argmin: the ray has its negative sample inside the mask), so the rest of the row gets out
        // of the way unevaluated.  Not if the first candidate is sample 0 - its bracket partner is the LAST sample
        // (ray_tracing.py:245-246): then, and when all four quarters are in, the row is completed and decided as a whole.
        const int ns = tp.n_steps, cw = coarse_window(ns);
        const int wn = (fl >> F_WIN_SHIFT) & F_WIN_MASK;
        const int have = wn * cw < ns ? wn * cw : ns;
        float *v = P.s.big + (size_t)r * ns;
        int i0 = -1, i1 = -1;
        for (int i = 0; i < have; ++i) {
            const float x = v[i];
            if (x < P.tau && i0 < 0) i0 = i;
            if (x < -P.tau && i1 < 0) i1 = i;
        }
        if (i1 >= 0 && i0 > 0) {
            for (int i = have; i < ns; ++i) v[i] = 3.0e38f;
            fl &= ~(F_WIN_MASK << F_WIN_SHIFT);                 // decided below, this round
        } else if (have < ns) {
            const bool rest = i0 == 0;                          // all remaining quarters at once
            nc = rest ? 4 - wn : 1;
            cwin = wn;
            fl = (fl & ~(F_WIN_MASK << F_WIN_SHIFT)) | ((rest ? 4 : wn + 1) << F_WIN_SHIFT);
            dense_which = 0;
            P.s.flags[r] = fl;
            ph = -1;
        } else {
            fl &= ~(F_WIN_MASK << F_WIN_SHIFT);                 // the whole row is in
        }
    }

    if (valid && ph == PH_SAMPLER_C) {
        // The n_steps samples hold COARSE values v16 with |v16 - v| < tau.  What the exact stage below decides from them:
        // the first negative sample `ind`, the signs at ind and ind-1 (bracket), and - unless the ray surely has a
        // negative sample and lies inside the object mask - the argmin.  Samples that cannot decide from their coarse
        // value are re-evaluated in split precision (refine list; next round's exact stage then sees exact values
        // exactly where it matters); with none, the exact stage runs right away on the coarse values.
        const int ns = tp.n_steps;
        const float tau = P.tau;
        const float *v = P.s.big + (size_t)r * ns;
        int i0 = -1, i1 = -1;           // first sample that may be negative / that surely is
        float vmin = v[0];
        for (int i = 0; i < ns; ++i) {
            const float x = v[i];
            if (x < tau && i0 < 0) i0 = i;
            if (x < -tau && i1 < 0) i1 = i;
            vmin = x < vmin ? x : vmin;
        }
        const bool obj = P.obj[r] != 0;
        const bool need_argmin = P.miss_argmin && !(obj && i1 >= 0);
        int n_sign = 0, n_min = 0;
        if (i0 >= 0) {
            const int end = i1 >= 0 ? i1 : ns;
            for (int i = i0; i < end; ++i)
                if (v[i] < tau) {
                    cmask[i >> 5] |= 1u << (i & 31);
                    ++n_sign;
                }
            // bracket quirk: ind = 0 pairs with sample ns-1 (ray_tracing.py:245-246)
            if (i0 == 0 && fabsf(v[ns - 1]) < tau && !((cmask[(ns - 1) >> 5] >> ((ns - 1) & 31)) & 1u)) {
                cmask[(ns - 1) >> 5] |= 1u << ((ns - 1) & 31);
                ++n_sign;
            }
        }
        if (need_argmin) {
            const float lim = vmin + 2.f * tau;
            for (int i = 0; i < ns; ++i)
                if (v[i] <= lim) {
                    cmask[i >> 5] |= 1u << (i & 31);
                    ++n_min;
                }
        }
        if (n_sign == 0 && n_min <= 1) {
            ph = PH_SAMPLER;            // every decision is certain from the coarse values
        } else {
            const int k = __popc(cmask[0]) + __popc(cmask[1]) + __popc(cmask[2]) + __popc(cmask[3]);
            if (k <= P.cap) n_ref = k; else qd = true;      // too many: all n_steps samples in split precision
            dense_which = 0;
            fl = (fl & ~F_PHASE) | PH_SAMPLER;
            P.s.flags[r] = fl;
            ph = -1;
        }
    }

    if (valid && ph == PH_SAMPLER) {
        // first sign change among the n_steps samples, argmin fallback, bracket (ray_tracing.py:203-255)
        const int ns = tp.n_steps;
        const float a = P.s.t_s[r], rng = fsub(P.s.t_e[r], a);
        const float *v = P.s.big + (size_t)r * ns;
        int ind = -1, zero = -1, amin = 0;
        float vmin = v[0];
        for (int i = 0; i < ns; ++i) {
            const float x = v[i];
            if (x < 0.f && ind < 0) ind = i;
            if (x == 0.f && zero < 0) zero = i;
            if (x < vmin) {
                vmin = x;
                amin = i;
            }
        }
        if (ind < 0) ind = zero >= 0 ? zero : ns - 1;
        const bool net_hit = v[ind] < 0.f;
        const bool obj = P.obj[r] != 0;
        float dist = fadd(a, fmul(P.lin[ind], rng));
        if (!(obj && net_hit)) dist = fadd(a, fmul(P.lin[amin], rng));
        fl = (fl & ~F_HIT) | (net_hit ? F_HIT : 0);
        const bool root = tp.training ? (net_hit && obj) : net_hit;
        bool go = false;
        if (root) {
            const int im = ind > 0 ? ind - 1 : ns - 1;
            const float hi = fadd(a, fmul(P.lin[ind], rng)), f_hi = v[ind];
            const float lo = fadd(a, fmul(P.lin[im], rng)), f_lo = v[im];
            const float mid = fmul(fadd(lo, hi), 0.5f);
            const bool work = (f_lo > 0.f) && (f_hi < 0.f) && (hi > lo);
            dist = mid;
            if (work && tp.n_rootfind_steps > 0) {
                P.s.lo[r] = lo;
                P.s.hi[r] = hi;
                P.s.mid[r] = mid;
                fl = (fl & ~(F_PHASE | (F_IT_MASK << F_IT_SHIFT))) | PH_BISECT;
                qt = true;
                go = true;
            }
        }
        if (go) {
            P.s.flags[r] = fl;
            ph = -1;
        } else {
            P.s.mid[r] = dist;
            ph = PH_POST;
        }
    }

    if (valid && ph == PH_BISECT) {
        // up to `bisect_levels` bisection steps per round (ray_tracing.py:264-277, per ray): the nodes of the next
        // levels of the bisection tree were evaluated speculatively last round; walk them with the sequential rule
        float lo = P.s.lo[r], hi = P.s.hi[r], mid = P.s.mid[r];
        const float *f = P.s.big + (size_t)r * tp.n_steps;
        int it = (fl >> F_IT_SHIFT) & F_IT_MASK;
        int node = 0;
        bool more = true;
        for (int level = 0; level < P.levels && more; ++level) {
            const float f_mid = f[node];
            ++consumed;
            const int bit = f_mid > 0.f ? 1 : 0;
            if (bit) lo = mid; else hi = mid;
            mid = fmul(fadd(lo, hi), 0.5f);
            ++it;
            more = (fsub(hi, lo) > 1e-6f) && (it < tp.n_rootfind_steps);
            node = 2 * node + 1 + bit;
        }
        P.s.mid[r] = mid;
        if (more) {
            P.s.lo[r] = lo;
            P.s.hi[r] = hi;
            fl = (fl & ~(F_IT_MASK << F_IT_SHIFT)) | (it << F_IT_SHIFT);
            P.s.flags[r] = fl;
            qt = true;
            ph = -1;
        } else {
            ph = PH_POST;
        }
    }

    if (valid && ph == PH_POST) {
        // after tracing / sampler: eval mode returns; training mode handles rays that miss (:71-97)
        float dist = P.s.mid[r];
        const bool hit = fl & F_HIT, samp = fl & F_SAMP, sph = fl & F_SPH;
        const bool obj = P.obj[r] != 0;
        bool done = true;
        if (tp.training) {
            const bool in_m = !hit && obj && !samp, out_m = !obj && !samp;
            if (in_m || out_m) {
                if (!sph) {
                    const float ox = P.o[r * 3], oy = P.o[r * 3 + 1], oz = P.o[r * 3 + 2];
                    const float dx = P.d[r * 3], dy = P.d[r * 3 + 1], dz = P.d[r * 3 + 2];
                    dist = -fadd(fadd(fmul(dx, ox), fmul(dy, oy)), fmul(dz, oz));
                } else {
                    if (hit && out_m) P.s.t_min[r] = dist;
                    fl = (fl & ~F_PHASE) | (coarse ? PH_MINSDF_C : PH_MINSDF);
                    qd = !coarse;
                    nc = coarse ? 4 : 0;
                    if (coarse && P.lip > 0.f) {
                        // staged search: a quarter row's worth of the depths, spread over their sorted order, first; whatever
                        // no evaluator writes stays +inf: neither a minimum nor within any band of one
                        float *v = P.s.big + (size_t)r * tp.n_steps;
                        for (int i = 0; i < tp.n_steps; ++i) v[i] = __builtin_inff();
                        fl = (fl & ~(F_WIN_MASK << F_WIN_SHIFT)) | (2 << F_WIN_SHIFT);
                        nc = 1;
                        cwin = CWIN_STAGE1;
                    }
                    P.s.flags[r] = fl;
                    n_alg = 1;
                    dense_which = 1;
                    done = false;
                }
            }
        }
        if (done) {
            finish(P, r, dist, hit);
            P.s.flags[r] = (fl & ~F_PHASE) | PH_DONE;
        }
        ph = -1;
    }

    if (valid && ph == PH_MINSDF_C && ((fl >> F_WIN_SHIFT) & F_WIN_MASK) >= 2) {
        // staged search (nefii_tracer_params.minsdf_lipschitz).  Walk the depths in sorted order between the first stage's
        // ones: lower bound of depth s between evaluated neighbours a < s < b from the Lipschitz bound L and the coarse
        // values c (|c - v| < tau):  v_s >= max(c_a - L (t_s - t_a), c_b - L (t_b - t_s)) - tau.
        //   stage 1 done (2): s is skipped for good when that bound exceeds best + tau >= the exact value at the lowest
        //     first-stage depth - it is not the argmin; the others go to the coarse evaluator one by one;
        //   stage 2 done (3): each of those is audited against the bound that kept it (c_s > bound - tau must hold).
        const int stage = (fl >> F_WIN_SHIFT) & F_WIN_MASK;
        const int ns = tp.n_steps, n1 = stage1_count(ns);
        float *v = P.s.big + (size_t)r * ns;
        const unsigned char *ord = P.s.ord + (size_t)minsdf_row(P, r) * ns;
        const float Llen = fmul(P.lip, fsub(P.s.t_max[r], P.s.t_min[r]));
        float best = __builtin_inff();
        for (int j = 0; j < n1; ++j) best = fminf(best, v[ord[stage1_pos(ns, j)]]);
        const float lim = fadd(best, P.tau);
        int ja = 0, k = 0, n_near = 0;
        float worst = 0.f;
        int probe = -1;             // one of the SKIPPED depths, picked by a hash of (ray, position): evaluated after all, so that
        unsigned probe_h = ~0u;     // the audit also sees the bound where it was relied upon (a ray costs one evaluation more)
        for (int kk = 1; kk < ns - 1; ++kk) {
            if (kk == stage1_pos(ns, ja + 1)) {
                ++ja;
                continue;
            }
            const int ia = ord[stage1_pos(ns, ja)], ib = ord[stage1_pos(ns, ja + 1)], is = ord[kk];
            const float sa = minsdf_step(P, r, ia), sb = minsdf_step(P, r, ib), ss = minsdf_step(P, r, is);
            const float lb = fsub(fmaxf(fsub(v[ia], fmul(Llen, fsub(ss, sa))), fsub(v[ib], fmul(Llen, fsub(sb, ss)))), P.tau);
            if (stage == 2) {
                if (!(fsub(lb, 1e-6f) > lim)) {
                    cmask[is >> 5] |= 1u << (is & 31);
                    ++k;
                } else if (n_near < NEAR_PROBES && !(fsub(lb, 1e-6f) > fadd(lim, fmul(2.f, P.tau)))) {
                    cmask[is >> 5] |= 1u << (is & 31);      // a probe of the audit: skipped by less than 2 tau (see the bracket search)
                    ++k, ++n_near;
                } else {
                    const unsigned h = ((unsigned)r * 2654435761u) ^ ((unsigned)(kk + 1) * 0x9E3779B1u);
                    const unsigned hh = (h ^ (h >> 15)) * 0x85EBCA6Bu;
                    if (hh < probe_h) probe_h = hh, probe = is;
                }
            } else if (v[is] < __builtin_inff()) {
                worst = fmaxf(worst, fsub(fsub(lb, P.tau), v[is]));
            }
        }
        fl &= ~(F_WIN_MASK << F_WIN_SHIFT);
        if (stage == 2 && probe >= 0) {
            cmask[probe >> 5] |= 1u << (probe & 31);
            ++k, ++n_near;
        }
        if (stage == 2 && n_near > 0) atomicAdd(P.counters + round * NCNT + 13, n_near);
        if (stage == 2 && k > 0) {
            if (k <= CREF_CAP) {
                n_ref = k;
                ref_coarse = true;
                fl |= 3 << F_WIN_SHIFT;
            } else {        // too many to list: the whole row, as without the staging
                cmask[0] = cmask[1] = cmask[2] = cmask[3] = 0u;
                nc = 4;
                dense_which = 1;
            }
            P.s.flags[r] = fl;
            ph = -1;
        } else {
            if (worst > 0.f) atomicMax(P.counters + round * NCNT + 12, __float_as_int(worst));
            cmask[0] = cmask[1] = cmask[2] = cmask[3] = 0u;      // the row's coarse values are in: go on below
        }
    }

    if (valid && ph == PH_MINSDF_C && ((fl >> F_WIN_SHIFT) & F_WIN_MASK) == 1) {
        // second stage of the two-stage refinement below: sample a (kept in the iteration bits) now holds its EXACT value
        // v*.  The exact argmin m has exact_m <= v*, hence coarse_m <= v* + tau: only such samples are refined; every other
        // one has exact > v* and keeps a coarse value > v* + tau - the exact stage's argmin over the mixed row is the
        // reference's (first index of the exact minimum).
        const int ns = tp.n_steps;
        const float *v = P.s.big + (size_t)r * ns;
        const int a = (fl >> F_IT_SHIFT) & F_IT_MASK;
        const float lim = v[a] + P.tau;
        int k = 0;
        for (int i = 0; i < ns; ++i)
            if (i != a && v[i] <= lim) {
                cmask[i >> 5] |= 1u << (i & 31);
                ++k;
            }
        fl &= ~((F_WIN_MASK << F_WIN_SHIFT) | (F_IT_MASK << F_IT_SHIFT));
        if (k == 0) {
            ph = PH_MINSDF;
        } else {
            if (k <= P.cap) n_ref = k; else qd = true;
            dense_which = 1;
            fl = (fl & ~F_PHASE) | PH_MINSDF;
            P.s.flags[r] = fl;
            ph = -1;
        }
    }

    if (valid && ph == PH_MINSDF_C) {
        // argmin over coarse values: every sample within 2 tau of the coarse minimum could be the exact one
        const int ns = tp.n_steps;
        const float *v = P.s.big + (size_t)r * ns;
        float vmin = v[0];
        int amin = 0;
        for (int i = 1; i < ns; ++i)
            if (v[i] < vmin) {
                vmin = v[i];
                amin = i;
            }
        const float lim = vmin + 2.f * P.tau;
        int k = 0;
        for (int i = 0; i < ns; ++i)
            if (v[i] <= lim) {
                cmask[i >> 5] |= 1u << (i & 31);
                ++k;
            }
        if (k <= 1) {
            ph = PH_MINSDF;
        } else if ((P.window & 2) && k >= 4 && ns <= 256) {
            // two stages: the coarse argmin alone first - its exact value v* bounds the exact minimum from above, and the
            // second stage's window (coarse <= v* + tau) is about half of this one's (coarse <= coarse min + 2 tau)
            cmask[0] = cmask[1] = cmask[2] = cmask[3] = 0u;
            cmask[amin >> 5] = 1u << (amin & 31);
            n_ref = 1;
            fl = (fl & ~((F_WIN_MASK << F_WIN_SHIFT) | (F_IT_MASK << F_IT_SHIFT))) | (1 << F_WIN_SHIFT) | (amin << F_IT_SHIFT);
            P.s.flags[r] = fl;
            ph = -1;
        } else {
            if (k <= P.cap) n_ref = k; else qd = true;
            dense_which = 1;
            fl = (fl & ~F_PHASE) | PH_MINSDF;
            P.s.flags[r] = fl;
            ph = -1;
        }
    }

    if (valid && ph == PH_MINSDF) {
        const int ns = tp.n_steps;
        const float *v = P.s.big + (size_t)r * ns;
        int amin = 0;
        float vmin = v[0];
        for (int i = 1; i < ns; ++i)
            if (v[i] < vmin) {
                vmin = v[i];
                amin = i;
            }
        const float tmin = P.s.t_min[r], tmax = P.s.t_max[r];
        const float dist = fadd(fmul(minsdf_step(P, r, amin), fsub(tmax, tmin)), tmin);
        finish(P, r, dist, fl & F_HIT);
        P.s.flags[r] = (fl & ~F_PHASE) | PH_DONE;
    }

    append_queries(P, round, qs && !cs, qe && !ce, qt, qd, nc, cwin, (unsigned)r, dense_which, consumed, n_alg, n_ref, cmask,
                   qs && cs, qe && ce, n_rep, ref_coarse);
    const int alive = __syncthreads_or(valid && (P.s.flags[r] & F_PHASE) != PH_DONE);
    if (threadIdx.x == 0) P.s.block_live[blockIdx.x] = alive;
}

// ---- SDF evaluation of one round's work list -------------------------------------------------
// queries of a round in split precision, in this order: singles | dense rays x n_steps | bisecting rays x tree nodes |
// refined coarse samples
struct RoundWork {
    int n_single;
    int64_t n_sd, n_sdt, total;
};
__device__ __forceinline__ RoundWork round_work(const Params &P, int round) {
    const int *c = P.counters + round * NCNT;
    RoundWork w;
    w.n_single = c[0];
    w.n_sd = (int64_t)c[0] + (int64_t)c[1] * P.p.n_steps;
    w.n_sdt = w.n_sd + (int64_t)c[2] * P.tri_nodes;
    w.total = w.n_sdt + c[4];
    return w;
}

// depth of dense sample i of ray r: the bracket search's linspace (ray_tracing.py:205) or the min-SDF search's uniform
// draws (:319); the one expression both the dense and the refine path use
__device__ __forceinline__ float dense_depth(const Params &P, int64_t r, int i, bool minsdf) {
    if (minsdf) {
        const float tmin = P.s.t_min[r], tmax = P.s.t_max[r];
        return fadd(fmul(minsdf_step(P, r, i), fsub(tmax, tmin)), tmin);
    }
    const float a = P.s.t_s[r];
    return fadd(a, fmul(P.lin[i], fsub(P.s.t_e[r], a)));
}

// decode the ROWS queries of a tile into points (raw[ROWS][9]) and result addresses (dest[ROWS])
// `old` (optional, [ROWS]): for a coarse-pass sample that is being re-evaluated, its coarse value (NaN for every other query) -
// the evaluator compares it with the split-precision value it is about to store (the online audit of coarse_tau)
template <int ROWS>
__device__ __forceinline__ void decode_tile(const Params &P, int64_t tile, const RoundWork &W, float *raw, float **dest,
                                            float *old = nullptr) {
    const int tid = threadIdx.x;
    if (tid >= ROWS) return;
    const int ns = P.p.n_steps;
    int64_t q = tile * ROWS + tid;
    float *dst = nullptr;
    float px = 0.f, py = 0.f, pz = 0.f;
    float coarse_v = __builtin_nanf("");
    if (q < W.total) {
        int64_t r;
        float t;
        if (q < W.n_single) {
            const unsigned e = P.s.singles[q];
            r = e >> 2;
            const int kind = e & 3;
            t = kind == Q_START ? P.s.t_s[r] : (kind == Q_END ? P.s.t_e[r] : P.s.mid[r]);
            dst = kind == Q_END ? &P.s.res_e[r] : &P.s.res_s[r];
            // a sphere-tracing query repeated in split precision: res_x still holds its coarse value
            if (old && P.tier_band > 0.f && (P.s.flags[r] & (kind == Q_END ? F_AUD_E : F_AUD_S))) coarse_v = *dst;
        } else if (q >= W.n_sdt) {
            const unsigned e = P.s.refine[q - W.n_sdt];
            r = e >> 7;
            const int i = e & 127;
            const int ph = P.s.flags[r] & F_PHASE;
            t = dense_depth(P, r, i, ph == PH_MINSDF || ph == PH_MINSDF_C);    // (_C: the first of the two refinement stages)
            dst = &P.s.big[(size_t)r * ns + i];
            if (old && ph != PH_SAMPLER_X) coarse_v = *dst;     // (the leading samples of a bracket search have no coarse value)
        } else if (q >= W.n_sd) {
            const int64_t qq = q - W.n_sd;
            const int64_t ti = qq / P.tri_nodes;
            const int j = (int)(qq - ti * P.tri_nodes);
            r = P.s.tri[ti];
            t = tri_depth(P.s.lo[r], P.s.hi[r], j);
            dst = &P.s.big[(size_t)r * ns + j];
        } else {
            const int64_t qq = q - W.n_single;
            const int64_t di = qq / ns;
            const int i = (int)(qq - di * ns);
            const unsigned e = P.s.dense[di];
            r = e >> 1;
            t = dense_depth(P, r, i, e & 1);
            dst = &P.s.big[(size_t)r * ns + i];
        }
        px = fadd(P.o[r * 3], fmul(t, P.d[r * 3]));
        py = fadd(P.o[r * 3 + 1], fmul(t, P.d[r * 3 + 1]));
        pz = fadd(P.o[r * 3 + 2], fmul(t, P.d[r * 3 + 2]));
    }
    dest[tid] = dst;
    if (old) old[tid] = coarse_v;
    float *rw = raw + tid * 9;
    rw[0] = px, rw[1] = py, rw[2] = pz;
    rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
}

// the same for the coarse evaluator's list: rays x n_steps samples
// queries [0, n_rows): the quarter rows' samples; [n_rows, total): the sphere-tracing queries of the tier (csingles)
template <int ROWS>
__device__ __forceinline__ void decode_tile_coarse(const Params &P, int round, int64_t tile, int64_t n_rows, int64_t total,
                                                   float *raw, float **dest, int tid = threadIdx.x) {
    if (tid >= ROWS) return;
    const int ns = P.p.n_steps, cw = coarse_window(ns);
    const int64_t q = tile * ROWS + tid;
    float *dst = nullptr;
    float px = 0.f, py = 0.f, pz = 0.f;
    const int64_t di = q / cw;
    const unsigned e = q < n_rows ? P.s.cdense[di] : 0u;
    int i = (int)(e >> 29) * cw + (int)(q - di * cw);
    if (q < n_rows && (e >> 29) == CWIN_STAGE1) {     // first stage of a staged min-SDF search: slot j -> sorted position -> sample
        const int j = (int)(q - di * cw);
        const int64_t r = (e & 0x1FFFFFFFu) >> 1;
        i = j >= stage1_count(ns) ? ns : (e & 1) ? P.s.ord[(size_t)minsdf_row(P, r) * ns + stage1_pos(ns, j)] : stage1_pos(ns, j);
    }
    const int64_t n_cs = P.counters[round * NCNT + 9];
    if (q >= n_rows + n_cs && q < total) {            // second stage: single depths
        const unsigned s = P.s.crefine[q - n_rows - n_cs];
        const int64_t r = s >> 7;
        const int si = (int)(s & 127u);
        const float t = dense_depth(P, r, si, (P.s.flags[r] & F_PHASE) == PH_MINSDF_C);      // (else: a bracket search's sample)
        dst = &P.s.big[(size_t)r * ns + si];
        px = fadd(P.o[r * 3], fmul(t, P.d[r * 3]));
        py = fadd(P.o[r * 3 + 1], fmul(t, P.d[r * 3 + 1]));
        pz = fadd(P.o[r * 3 + 2], fmul(t, P.d[r * 3 + 2]));
    } else if (q >= n_rows && q < total) {
        const unsigned s = P.s.csingles[q - n_rows];
        const int64_t r = s >> 2;
        const bool end = (s & 3) == Q_END;
        const float t = end ? P.s.t_e[r] : P.s.t_s[r];
        dst = end ? &P.s.res_e[r] : &P.s.res_s[r];
        px = fadd(P.o[r * 3], fmul(t, P.d[r * 3]));
        py = fadd(P.o[r * 3 + 1], fmul(t, P.d[r * 3 + 1]));
        pz = fadd(P.o[r * 3 + 2], fmul(t, P.d[r * 3 + 2]));
    } else if (q < n_rows && i < ns) {
        const int64_t r = (e & 0x1FFFFFFFu) >> 1;
        const float t = dense_depth(P, r, i, e & 1);
        dst = &P.s.big[(size_t)r * ns + i];
        px = fadd(P.o[r * 3], fmul(t, P.d[r * 3]));
        py = fadd(P.o[r * 3 + 1], fmul(t, P.d[r * 3 + 1]));
        pz = fadd(P.o[r * 3 + 2], fmul(t, P.d[r * 3 + 2]));
    }
    dest[tid] = dst;
    float *rw = raw + tid * 9;
    rw[0] = px, rw[1] = py, rw[2] = pz;
    rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
}

__global__ __launch_bounds__(256, 1) void eval_kernel(Params P, nefii_mlp m, int round) {
    NEFII_CLAIM_SIMD_1();
    __shared__ Lds lds;
    __shared__ float raw[TILE * 9];
    __shared__ float *dest[TILE];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    const int64_t n_tiles = (total + TILE - 1) / TILE;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    const int Lm1 = m.n_layers - 1;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        decode_tile<TILE>(P, tile, W, raw, dest);
        __syncthreads();
        encode_tile(m, raw, lds.E, ke);
        __syncthreads();
        for (int l = 0; l <= Lm1; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            int ntw;
            layer_gemm(L, lds.X, lds.E, L.w_fwd, L.n_pad >> 5, acc, ntw);
            __syncthreads();
            if (l < Lm1) {
                NEFII_ACT_SWITCH(m.act, { NEFII_FOR_ACC(acc, ntw, { lds.X[row * XS + col] = act_fwd(val + L.bias[col], ACT); }) })
            } else {
                NEFII_FOR_ACC(acc, ntw, {
                    if (col == 0 && dest[row]) *dest[row] = val + L.bias[0];
                })
            }
            __syncthreads();
        }
    }
}

// split-precision variant (3 x fp16 MFMA per k-step, mlp_tile.h)
__global__ __launch_bounds__(256, 1) void eval_kernel16(Params P, nefii_mlp m, int round) {    // 81 KB of LDS: one workgroup per CU anyway
    NEFII_CLAIM_SIMD_1();
    __shared__ Lds16 lds;
    __shared__ float raw[TILE * 9];
    __shared__ float *dest[TILE];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    const int64_t n_tiles = (total + TILE - 1) / TILE;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    const int Lm1 = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        decode_tile<TILE>(P, tile, W, raw, dest);
        __syncthreads();
        encode_tile16(m, raw, lds, ke);
        __syncthreads();
        for (int l = 0; l <= Lm1; ++l) {
            const nefii_layer &L = m.layer[l];
            f32x16 acc[4];
            int ntw;
            layer_gemm16(L, lds, L.n_pad >> 5, acc, ntw);
            __syncthreads();
            if (l < Lm1) {
                NEFII_ACT_SWITCH(m.act, {
                    NEFII_FOR_ACC(acc, ntw, {
                        const float hval = act_fwd(val * inv_scale + L.bias[col], ACT);
                        split16a(hval, lds.Xh[row * XS16 + col], lds.Xl[row * XS16 + col]);
                    })
                })
            } else {
                NEFII_FOR_ACC(acc, ntw, {
                    if (col == 0 && dest[row]) *dest[row] = val * inv_scale + L.bias[0];
                })
            }
            __syncthreads();
        }
    }
}

// wide split-precision variant: 64 queries per workgroup of 8 waves (mlp_tile.h, Lds16w)
// One tile: raw[64][9] holds the points, dest[64] where each SDF value goes (nullptr = padding row).
__device__ __forceinline__ void sdf_tile16w(const nefii_mlp &m, Lds16w &lds, const float *raw, float *const *dest,
                                            int ke) {
    const int Lm1 = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE);
    encode_tile16w(m, raw, lds, ke);
    __syncthreads();
    for (int l = 0; l <= Lm1; ++l) {
        const nefii_layer &L = m.layer[l];
        f32x16 acc[4];
        int nct;
        layer_gemm16w(L, lds, L.n_pad >> 5, acc, nct);
        __syncthreads();
        if (l < Lm1) {
            const float k16 = inv_scale * A16_SCALE;
            NEFII_ACT_SWITCH(m.act, {
                NEFII_FOR_ACC_WT(acc, nct, L.n_pad >> 5, {
                    const float4v b = *reinterpret_cast<const float4v *>(L.bias + f0);
                    float4v hs;       // A16_SCALE * activation
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {
                        const float zs = __builtin_fmaf(v[k], k16, b[k] * A16_SCALE);
                        hs[k] = ACT == NEFII_ACT_SOFTPLUS100 ? softplus100_s16(zs)
                                                             : act_fwd(zs * (1.f / A16_SCALE), ACT) * A16_SCALE;
                    }
                    const half4 hi = __builtin_convertvector(hs, half4);
                    const half4 lo = __builtin_convertvector(hs - __builtin_convertvector(hi, float4v), half4);
                    *reinterpret_cast<half4 *>(&lds.Xh[query * XS16 + f0]) = hi;
                    *reinterpret_cast<half4 *>(&lds.Xl[query * XS16 + f0]) = lo;
                })
            })
        } else {
            NEFII_FOR_ACC_WT(acc, nct, L.n_pad >> 5, {
                if (f0 == 0 && dest[query]) *dest[query] = v[0] * inv_scale + L.bias[0];
            })
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 2) void eval_kernel16w(Params P, nefii_mlp m, int round) {
    NEFII_CLAIM_SIMD_2();
    __shared__ Lds16w lds;
    __shared__ float raw[TILE_W * 9];
    __shared__ float *dest[TILE_W];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    const int64_t n_tiles = (total + TILE_W - 1) / TILE_W;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        decode_tile<TILE_W>(P, tile, W, raw, dest);
        __syncthreads();
        sdf_tile16w(m, lds, raw, dest, ke);
    }
}

// pipelined variant (mlp_tile.h "16p"): 512-wide hidden layers, fragment stream never drains.
// Two instantiations are launched per round and the round's query count (known on the device only) picks the one that
// works: RT = 2 (64-query tiles) for rounds of more than SMALL_ROUND queries, RT = 1 (32-query tiles) for the others - a
// round with fewer 64-query tiles than half the CUs puts twice as many CUs to work on tiles with half the matrix work
// and epilogue (the fragment stream per tile is the same: ~110 instead of ~150 us per round).  Kept as two kernels:
// fused into one, the register allocation of the 64-query path degraded (60 spills, dense rounds +25 %).
constexpr int64_t SMALL_ROUND = 64 * 128;
// nefii_tracer_params.small_round overrides the threshold: a trace that runs beside others is bound by chip time, not by
// its own latency, and a 32-query tile costs 1.5x the chip time per query of a 64-query one
__device__ __forceinline__ int64_t small_round(const Params &P) { return P.p.small_round > 0 ? P.p.small_round : SMALL_ROUND; }
template <int NW, int RT>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) void eval_kernel16p(Params P, nefii_mlp m, int round) {
    static_assert(NW == 8, "register claim below: two waves per SIMD");
    NEFII_CLAIM_SIMD_2();
    __shared__ Lds16p lds;
    __shared__ float raw[TILE_W * 9];
    __shared__ float *dest[TILE_W];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    if ((total <= small_round(P)) != (RT == 1)) return;
    constexpr int ROWS = 32 * RT;
    const int64_t n_tiles = (total + ROWS - 1) / ROWS;
    if (blockIdx.x >= n_tiles) return;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    typename P16<NW>::Stage b[P16<NW>::NB];
    PCursor cur;
    int ph = 0;
    prime16p<NW>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        decode_tile<ROWS>(P, tile, W, raw, dest);
        __syncthreads();
        sdf_tile16p<NW, RT>(m, lds, raw, dest, b, cur, ph, ke);
    }
}

// the padded stream of the deep-prefetch instance multiplies whatever follows a layer's own K columns in the activation
// image (the row's pad columns, the next row, `tail`) by zero weights: all of it must be finite from the first tile on
template <int FT>
__device__ __forceinline__ void zero_lds(LdsQ<FT> &lds) {
    uint32_t *p = reinterpret_cast<uint32_t *>(&lds);
    for (int i = threadIdx.x; i < (int)(sizeof(LdsQ<FT>) / 4); i += blockDim.x) p[i] = 0u;
    __syncthreads();
}

// the same on v_mfma_f32_16x16x32_f16 (mlp_tile.h "16q"; nefii_mlp.reserved == 1).  FT = 4: 512-wide hidden layers,
// 64- / 32-query tiles; FT = 2: 256-wide hidden layers (conf_neus.conf), 96- / 32-query tiles.
template <int QT, int FT, bool DEEP = false>
__global__ __launch_bounds__(512, 2) void eval_kernel16q(Params P, nefii_mlp m, int round) {
    NEFII_CLAIM_SIMD_2();
    constexpr int RMAX = QGeo<FT>::ROWS;
    __shared__ LdsQ<FT> lds;
    __shared__ float raw[RMAX * 9];
    __shared__ float *dest[RMAX];
    __shared__ float old[RMAX];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    // Which instance takes which queries.  Rounds up to SMALL_ROUND: all in 32-query tiles.  Larger rounds: big tiles
    // (64 / 96 queries); when the big tiles form whole waves of one tile per CU plus a remainder that fits one wave of
    // 32-query tiles, that remainder goes to the 32-query instance (a wave of those is done in ~110 instead of ~160 us).
    constexpr int ROWS = 16 * QT, BIG = QGeo<FT>::ROWS, NCU = 256;
    int64_t first = 0, n_tiles;                     // this instance's tiles: first .. first + n_tiles - 1, ROWS queries each
    if (total <= small_round(P)) {
        n_tiles = QT == 2 ? (total + 31) / 32 : 0;
    } else {
        const int64_t nbig = (total + BIG - 1) / BIG, whole = nbig / NCU * NCU, rem = nbig - whole;
        const bool split = whole > 0 && rem > 0 && rem * (BIG / 32) <= NCU;
        if (QT == 2) {
            first = split ? whole * (BIG / 32) : 0;
            n_tiles = split ? (total - whole * BIG + 31) / 32 : 0;
        } else {
            n_tiles = split ? whole : nbig;
        }
    }
    if (blockIdx.x >= n_tiles) return;
    // DEEP: 8 fragment stages instead of 4 for the 32-query instance of the 512-wide shape.  It pays when few CUs stream
    // (batches of <= 1024 rays: <= 64 tiles per round, each bound by the latency of its own 7.6 MB stream - config 1:
    // 2.49 -> 2.17 ms per step); with every CU streaming the tiles are bound by L2 bandwidth instead and the deeper
    // pipeline only adds its priming (config 2's small rounds: 104 -> 110 us), so the host picks it by batch size.
    constexpr int NB = DEEP ? 8 : 4;
    static_assert(!DEEP || (QT == 2 && FT == 4), "deep prefetch: the 32-query instance of the 512-wide shape");
    if (NB == 8) zero_lds(lds);
    P16<8>::Stage b[NB];
    PCursor cur;
    prime16q<FT, NB>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const bool audit = P.tau > 0.f;
        decode_tile<ROWS>(P, first + tile, W, raw, dest, audit ? old : nullptr);
        __syncthreads();
        sdf_tile16q<QT, FT, NB>(m, lds, raw, dest, b, cur, audit ? old : nullptr, P.counters + round * NCNT + 8);
    }
}

// "16f" (mlp_tile.h, nefii_tracer_params.split_fp8): the same work lists and tile shapes as eval_kernel16q<QT, 4>, the correction
// products on block-scaled fp8; f8_stream = the fifth copy of nefii_mlp.w_stream
template <typename LDS>
__device__ __forceinline__ void zero_lds_f(LDS &lds) {
    uint32_t *p = reinterpret_cast<uint32_t *>(&lds);
    for (int i = threadIdx.x; i < (int)(sizeof(LDS) / 4); i += blockDim.x) p[i] = 0u;
    __syncthreads();
}
template <int QT>
__global__ __launch_bounds__(512, 2) void eval_kernel16f(Params P, nefii_mlp m, int round, const void *f8_stream) {
    NEFII_CLAIM_SIMD_2();
    constexpr int RMAX = QGeo<4>::ROWS;
    __shared__ LdsF lds;
    __shared__ float raw[RMAX * 9];
    __shared__ float *dest[RMAX];
    __shared__ float old[RMAX];
    const RoundWork W = round_work(P, round);
    const int64_t total = W.total;
    constexpr int ROWS = 16 * QT, BIG = RMAX, NCU = 256;      // (which instance takes which queries: as eval_kernel16q)
    int64_t first = 0, n_tiles;
    if (total <= small_round(P)) {
        n_tiles = QT == 2 ? (total + 31) / 32 : 0;
    } else {
        const int64_t nbig = (total + BIG - 1) / BIG, whole = nbig / NCU * NCU, rem = nbig - whole;
        const bool split = whole > 0 && rem > 0 && rem * (BIG / 32) <= NCU;
        if (QT == 2) {
            first = split ? whole * (BIG / 32) : 0;
            n_tiles = split ? (total - whole * BIG + 31) / 32 : 0;
        } else {
            n_tiles = split ? whole : nbig;
        }
    }
    if (blockIdx.x >= n_tiles) return;
    zero_lds_f(lds);        // the 128-padded stream multiplies what follows a layer's own columns by zero weights: keep it finite
    SStage<4> b[4];
    PCursor cur;
    prime16f(m, f8_stream, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const bool audit = P.tau > 0.f;
        decode_tile<ROWS>(P, first + tile, W, raw, dest, audit ? old : nullptr);
        __syncthreads();
        sdf_tile16f<QT>(m, lds, raw, dest, b, cur, audit ? old : nullptr, P.counters + round * NCNT + 8);
    }
}

__global__ __launch_bounds__(512, 2) void sdf_points_kernel16f(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ out, const void *f8_stream) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = QGeo<4>::ROWS, QT = ROWS / 16;
    __shared__ LdsF lds;
    __shared__ float raw[ROWS * 9];
    __shared__ float *dest[ROWS];
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    zero_lds_f(lds);
    SStage<4> b[4];
    PCursor cur;
    prime16f(m, f8_stream, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tid = threadIdx.x;
        if (tid < ROWS) {
            const int64_t q = tile * ROWS + tid;
            float *rw = raw + tid * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            dest[tid] = live ? out + q : nullptr;
        }
        __syncthreads();
        sdf_tile16f<QT>(m, lds, raw, dest, b, cur);
    }
}

// the coarse evaluator (mlp_tile.h "16s") over the round's coarse list: one fp16 pass, 16 * QT queries per tile
template <int FT, int ROWS>
__device__ __forceinline__ void zero_lds_s(LdsS<FT, ROWS> &lds) {
    uint32_t *p = reinterpret_cast<uint32_t *>(&lds);
    for (int i = threadIdx.x; i < (int)(sizeof(LdsS<FT, ROWS>) / 4); i += blockDim.x) p[i] = 0u;
    __syncthreads();
}

// DB (the 64- / 96-row default tiles): two activation images, one barrier per layer (mlp_tile.h, sdf_tile16s2)
template <int FT, int ROWS, bool DB>
using LdsSx = typename std::conditional<DB, LdsS2<FT, ROWS>, LdsS<FT, ROWS>>::type;
template <typename LDS>
__device__ __forceinline__ void zero_lds_any(LDS &lds) {
    uint32_t *p = reinterpret_cast<uint32_t *>(&lds);
    for (int i = threadIdx.x; i < (int)(sizeof(LDS) / 4); i += blockDim.x) p[i] = 0u;
    __syncthreads();
}

template <int QT, int FT, bool DB = true>
__global__ __launch_bounds__(512, 2) void eval_kernel16s(Params P, nefii_mlp m, int round) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = 16 * QT, RMAX = ROWS;
    __shared__ LdsSx<FT, ROWS, DB> lds;
    __shared__ float raw[RMAX * 9];
    __shared__ float *dest[RMAX];
    const int64_t n_rows = (int64_t)P.counters[round * NCNT + 5] * coarse_window(P.p.n_steps);
    const int64_t total = n_rows + P.counters[round * NCNT + 9] + P.counters[round * NCNT + 11];
    const int64_t n_tiles = (total + ROWS - 1) / ROWS;
    if (blockIdx.x >= n_tiles) return;
    zero_lds_any(lds);      // the K-padded stream multiplies what follows a layer's own columns by zero weights: keep it finite
    SStage<FT> b[4];
    PCursor cur;
    prime16s<FT>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        decode_tile_coarse<ROWS>(P, round, tile, n_rows, total, raw, dest);
        __syncthreads();
        if constexpr (DB)
            sdf_tile16s2<QT, FT>(m, lds, raw, dest, b, cur);
        else
            sdf_tile16s<QT, FT, false>(m, lds, raw, dest, b, cur);
    }
}

template <int QT, int FT, bool DB = true>
__global__ __launch_bounds__(512, 2) void sdf_points_kernel16s(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ out) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = 16 * QT, RMAX = ROWS;
    __shared__ LdsSx<FT, ROWS, DB> lds;
    __shared__ float raw[RMAX * 9];
    __shared__ float *dest[RMAX];
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    zero_lds_any(lds);
    SStage<FT> b[4];
    PCursor cur;
    prime16s<FT>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tid = threadIdx.x;
        if (tid < ROWS) {
            const int64_t q = tile * ROWS + tid;
            float *rw = raw + tid * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            dest[tid] = live ? out + q : nullptr;
        }
        __syncthreads();
        if constexpr (DB)
            sdf_tile16s2<QT, FT>(m, lds, raw, dest, b, cur);
        else
            sdf_tile16s<QT, FT, false>(m, lds, raw, dest, b, cur);
#ifdef NEFII_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp_tile = g_stamp_tile + 1;
        __syncthreads();
#endif
    }
}

// "16d" (mlp_tile.h): the single-pass tile on four waves; a workgroup is two independent four-wave groups, each a persistent
// worker over its own tiles (tile = 2 x workgroup + group, stride 2 x grid), image, decode buffers and barrier counter
template <int QT>
__global__ __launch_bounds__(512, 2) void eval_kernel16d(Params P, nefii_mlp m, int round) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = 16 * QT;
    __shared__ LdsS2<4, ROWS> lds;
    __shared__ float raw[2 * ROWS * 9];
    __shared__ float *dest[2 * ROWS];
    __shared__ unsigned bar[3];
    const int64_t n_rows = (int64_t)P.counters[round * NCNT + 5] * coarse_window(P.p.n_steps);
    const int64_t total = n_rows + P.counters[round * NCNT + 9] + P.counters[round * NCNT + 11];
    const int64_t n_tiles = (total + ROWS - 1) / ROWS;
    if (blockIdx.x >= n_tiles) return;
    if (threadIdx.x < 3) bar[threadIdx.x] = 0u;
    zero_lds_any(lds);      // (ends with the one workgroup barrier of this kernel)
    const int g = threadIdx.x >> 8;
    // group 0 takes tiles [0, grid), group 1 [grid, 2 grid), ...: a round of no more tiles than workgroups runs one group per CU
    if (blockIdx.x + g * (int64_t)gridDim.x < n_tiles) {
        GroupBarrier gb(&bar[g]);
        SStage<4> b[4];
        PCursor cur[2];
        prime16d<QT>(m, b, cur);
        for (int64_t tile = blockIdx.x + g * (int64_t)gridDim.x; tile < n_tiles; tile += 2 * (int64_t)gridDim.x) {
            decode_tile_coarse<ROWS>(P, round, tile, n_rows, total, raw + g * ROWS * 9, dest + g * ROWS, threadIdx.x & 255);
            gb.sync();
            sdf_tile16d<QT>(m, lds.X[g], raw + g * ROWS * 9, dest + g * ROWS, b, cur, gb);
        }
    }
    GroupBarrier::hold(&bar[2]);
}

template <int QT>
__global__ __launch_bounds__(512, 2) void sdf_points_kernel16d(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ out) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = 16 * QT;
    __shared__ LdsS2<4, ROWS> lds;
    __shared__ float raw[2 * ROWS * 9];
    __shared__ float *dest[2 * ROWS];
    __shared__ unsigned bar[3];
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    if (threadIdx.x < 3) bar[threadIdx.x] = 0u;
    zero_lds_any(lds);
    const int g = threadIdx.x >> 8, tl = threadIdx.x & 255;
    if (blockIdx.x + g * (int64_t)gridDim.x >= n_tiles) {
        GroupBarrier::hold(&bar[2]);
        return;
    }
    GroupBarrier gb(&bar[g]);
    SStage<4> b[4];
    PCursor cur[2];
    prime16d<QT>(m, b, cur);
#ifdef NEFII_STAMPS
    if (tl == 0 && blockIdx.x < 256) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_dhwid[2 * (2 * blockIdx.x + g)] = hw, g_dhwid[2 * (2 * blockIdx.x + g) + 1] = xcc;
    }
#endif
    int seq = 0;
    float *rawg = raw + g * ROWS * 9;
    float **destg = dest + g * ROWS;
    for (int64_t tile = blockIdx.x + g * (int64_t)gridDim.x; tile < n_tiles; tile += 2 * (int64_t)gridDim.x, ++seq) {
        if (tl < ROWS) {
            const int64_t q = tile * ROWS + tl;
            float *rw = rawg + tl * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            destg[tl] = live ? out + q : nullptr;
        }
        gb.sync();
        sdf_tile16d<QT>(m, lds.X[g], rawg, destg, b, cur, gb, seq);
    }
    GroupBarrier::hold(&bar[2]);
}

template <int FT>
__global__ __launch_bounds__(512, 2) void sdf_points_kernel16q(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ out) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = QGeo<FT>::ROWS, QT = ROWS / 16;
    __shared__ LdsQ<FT> lds;
    __shared__ float raw[ROWS * 9];
    __shared__ float *dest[ROWS];
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    P16<8>::Stage b[4];
    PCursor cur;
    prime16q<FT>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tid = threadIdx.x;
        if (tid < ROWS) {
            const int64_t q = tile * ROWS + tid;
            float *rw = raw + tid * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            dest[tid] = live ? out + q : nullptr;
        }
        __syncthreads();
        sdf_tile16q<QT, FT>(m, lds, raw, dest, b, cur);
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) void sdf_points_kernel16p(nefii_mlp m,
                                                                                const float *__restrict__ x,
                                                                                int64_t n, float *__restrict__ out) {
    NEFII_CLAIM_SIMD_2();
    __shared__ Lds16p lds;
    __shared__ float raw[TILE_W * 9];
    __shared__ float *dest[TILE_W];
    const int64_t n_tiles = (n + TILE_W - 1) / TILE_W;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    typename P16<NW>::Stage b[P16<NW>::NB];
    PCursor cur;
    int ph = 0;
    prime16p<NW>(m, b, cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tid = threadIdx.x;
        if (tid < TILE_W) {
            const int64_t q = tile * TILE_W + tid;
            float *rw = raw + tid * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            dest[tid] = live ? out + q : nullptr;
        }
        __syncthreads();
        sdf_tile16p<NW, 2>(m, lds, raw, dest, b, cur, ph, ke);
#ifdef NEFII_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp_tile = g_stamp_tile + 1;
        __syncthreads();
#endif
    }
}

#ifdef NEFII_STAMPS
extern "C" int nefii_debug_dstamps(unsigned long long *host_out, unsigned *hwid_out) {
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dstamps), sizeof(unsigned long long) * 512 * 12 * 5);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyFromSymbol(hwid_out, HIP_SYMBOL(g_dhwid), sizeof(unsigned) * 1024);
    return (int)e;
}
extern "C" int nefii_debug_stamps(unsigned long long *host_out) {
    int zero = 0;
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 2 * 8 * 12 * 5);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_tile), &zero, sizeof(int));
    return (int)e;
}
#endif

constexpr int P16W = 8;      // waves per workgroup of the pipelined kernels
__device__ __forceinline__ int layer_units_dev(const nefii_layer &L, int ft) {
    return ft == 2 ? q_units<2>(L) : (L.k_x + L.k_e) >> 4;
}

// shapes the pipelined kernels take: every hidden layer W wide (512; 256 with the 16x16x32 layout only), inputs of 0 or
// W previous features plus 0 or 64 encoding columns, a W-deep last layer.  Returns the feature tiles per wave
// (W / 128), 0 when the shape does not fit.
int shape16p(const nefii_mlp *m) {
    const int NH = m->n_layers - 1;
    if (NH < 1) return 0;
    const int W = m->layer[0].n_pad;
    if (W != 512 && !(W == 256 && m->reserved == 1)) return 0;
    for (int l = 0; l < NH; ++l) {
        const nefii_layer &L = m->layer[l];
        if (L.n_pad != W || (L.k_x != 0 && L.k_x != W) || (L.k_e != 0 && L.k_e != 64) || L.k_x + L.k_e == 0) return 0;
    }
    const nefii_layer &Ll = m->layer[NH];
    return Ll.k_x == W && Ll.k_e == 0 ? W / 128 : 0;
}
int fits16p(const nefii_mlp *m) { return m->w_stream ? shape16p(m) : 0; }

int layer_units(const nefii_layer &L, int ft) { return ft == 2 ? q_units<2>(L) : (L.k_x + L.k_e) >> 4; }
int stream_steps(const nefii_mlp *m) {
    const int ft = shape16p(m);
    int G = 0;
    for (int l = 0; l < m->n_layers - 1; ++l) G += layer_units(m->layer[l], ft);
    return G;
}
// units of the second, K-padded copy the deep-prefetch 32-query instance reads (512-wide nets, 16x16x32 layout)
int stream_steps8(const nefii_mlp *m) {
    if (shape16p(m) != 4 || m->reserved != 1) return 0;
    int G = 0;
    for (int l = 0; l < m->n_layers - 1; ++l) G += q_units8(m->layer[l]);
    return G;
}

// layout 0 (32x32x16 fragments): dst[(wave*G + g)*256 + i] <- the 4 KiB fragment block of (k-step g of the layer
// sequence, column tiles 2 wave, 2 wave + 1), a straight copy.
// layout 1 (16x16x32 fragments, nefii_mlp.reserved == 1), 512-wide: unit g is a HALF step (32-deep k-step g/2 of the
// layer sequence, feature-tile pair g&1): dst[((wave*G + g)*4 + 2 f + part)*64 + lane][j] = W[n = 64 wave + 16 (2 (g&1)
// + f) + (lane&15)][k = 32 s32 + 8 (lane>>4) + j], gathered from the layer's 32x32x16 fragments (one source half8 per
// destination half8).  256-wide: unit g is a whole 32-deep k-step of the layer's K padded to a multiple of 128,
// n = 32 wave + 16 f + (lane&15); k-steps past the layer's own K hold zeros.
// Gw: units per wave of the destination copy (G, or more when other units follow in the same copy).
__global__ void pack_sdf_stream_kernel(nefii_mlp m, half8 *__restrict__ dst, int Gw, int ft, int padded) {
    const int g = blockIdx.x, wave = blockIdx.y, G = Gw;
    int l = 0, s = g;
    while (s >= (padded ? q_units8(m.layer[l]) : layer_units_dev(m.layer[l], ft)))
        s -= padded ? q_units8(m.layer[l]) : layer_units_dev(m.layer[l], ft), ++l;
    const half8 *w = reinterpret_cast<const half8 *>(m.layer[l].w_f16x3);
    if (m.reserved != 1) {
        const half8 *src = w + ((size_t)s * 16 + 2 * wave) * 2 * 64;
        dst[((size_t)wave * G + g) * 256 + threadIdx.x] = src[threadIdx.x];
        return;
    }
    const int frag = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = frag >> 1, part = frag & 1, kg = lane >> 4;
    if (ft == 2) {
        const int n = 32 * wave + 16 * f + (lane & 15);
        const int s16 = 2 * s + (kg >> 1), t = n >> 5, lane_src = (n & 31) + 32 * (kg & 1);
        half8 v;
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
        if (s16 < ((m.layer[l].k_x + m.layer[l].k_e) >> 4)) v = w[(((size_t)s16 * 8 + t) * 2 + part) * 64 + lane_src];
        dst[((size_t)wave * G + g) * 256 + threadIdx.x] = v;
        return;
    }
    const int half = s & 1, s32 = s >> 1;
    const int n = 64 * wave + 16 * (2 * half + f) + (lane & 15);
    const int s16 = 2 * s32 + (kg >> 1), t = n >> 5, lane_src = (n & 31) + 32 * (kg & 1);
    half8 v;
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;          // k-steps past the layer's own K (padded copy): zeros
    if (s16 < ((m.layer[l].k_x + m.layer[l].k_e) >> 4)) v = w[(((size_t)s16 * 16 + t) * 2 + part) * 64 + lane_src];
    dst[((size_t)wave * G + g) * 256 + threadIdx.x] = v;
}

// third copy (nefii_mlp.reserved == 1 only): the single-pass evaluator's stream.  Unit g of a wave = one 32-deep k-step
// of the layer sequence with every layer's K padded to a multiple of 128, hi fragments of the wave's FT feature tiles:
// dst[((wave*G + g)*FT + f)*64 + lane][j] = hi(W[n = 16 FT wave + 16 f + (lane&15)][k = 32 s + 8 (lane>>4) + j]),
// zeros past the layer's own K.  blockDim.x = 64 FT.
__global__ void pack_sdf_stream_sp_kernel(nefii_mlp m, half8 *__restrict__ dst, int G, int ft) {
    const int g = blockIdx.x, wave = blockIdx.y;
    int l = 0, s = g;
    while (s >= s_units(m.layer[l])) s -= s_units(m.layer[l]), ++l;
    const half8 *w = reinterpret_cast<const half8 *>(m.layer[l].w_f16x3);
    const int f = threadIdx.x >> 6, lane = threadIdx.x & 63, kg = lane >> 4;
    const int n = 16 * ft * wave + 16 * f + (lane & 15);
    const int s16 = 2 * s + (kg >> 1), t = n >> 5, lane_src = (n & 31) + 32 * (kg & 1);
    const int NT = m.layer[l].n_pad >> 5;
    half8 v;
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
    if (s16 < ((m.layer[l].k_x + m.layer[l].k_e) >> 4)) v = w[(((size_t)s16 * NT + t) * 2) * 64 + lane_src];
    dst[((size_t)wave * G + g) * ft * 64 + threadIdx.x] = v;
}

// units of the single-pass copy (0: the net has none)
int stream_steps_sp(const nefii_mlp *m) {
    if (!shape16p(m) || m->reserved != 1) return 0;
    int G = 0;
    for (int l = 0; l < m->n_layers - 1; ++l) G += s_units(m->layer[l]);
    return G;
}

// ---- fourth copy: the value + gradient kernel's stream (sdf_value_grad16q_kernel) ------------------------------------------
// Softplus nets of the pipelined shapes (512- or 256-wide) with transposed fragments on every layer.  Per wave: the
// forward units of the plain 16x16x32 stream, then the BACKWARD units - for l = NH-1 .. 1 the units of the transposed
// layer (contraction over layer l's W outputs, the wave's share of its W hidden inputs as output features), gathered
// from nefii_layer.w_bwd_f16x3 - so that one cursor runs forward, backward, and wraps to the next tile's forward.
// Returns the feature tiles per wave (4 / 2), 0 when the net does not take the kernel.
int vg_shape(const nefii_mlp *m) {
    const int ft = shape16p(m);
    if (!ft || m->reserved != 1 || m->act != NEFII_ACT_SOFTPLUS100) return 0;
    const int NH = m->n_layers - 1;
    if (NH < 2 || NH > 12 || m->layer[0].k_e != 64) return 0;
    if (m->enc_freqs[0] < 0 || m->enc_freqs[1] >= 0 || m->enc_freqs[2] >= 0 || m->feat_width != 0) return 0;
    for (int l = 0; l <= NH; ++l)
        if (!m->layer[l].w_f16x3 || !m->layer[l].w_bwd_f16x3 || !m->layer[l].bias) return 0;
    return ft;
}
// backward units of one layer: 16-deep half steps (512-wide), whole 32-deep steps (256-wide)
__host__ __device__ __forceinline__ int vg_units_layer(int n_pad, int ft) { return ft == 4 ? n_pad >> 4 : n_pad >> 5; }
int vg_units_bwd(const nefii_mlp *m) {
    const int ft = vg_shape(m);
    return ft ? (m->n_layers - 2) * vg_units_layer(m->layer[0].n_pad, ft) : 0;
}
// half8 offset of the copy in w_stream
size_t vg_stream_offset(const nefii_mlp *m) {
    return (size_t)8 * (stream_steps(m) + stream_steps8(m)) * 256 + (size_t)8 * stream_steps_sp(m) * shape16p(m) * 64;
}

__global__ void pack_sdf_stream_bwd_kernel(nefii_mlp m, half8 *__restrict__ dst, int Gw, int g0, int ft) {
    const int g = blockIdx.x, wave = blockIdx.y;
    const int NH = m.n_layers - 1, U = vg_units_layer(m.layer[0].n_pad, ft);
    const int l = NH - 1 - g / U, s = g % U;
    const nefii_layer &L = m.layer[l];
    const half8 *wb = reinterpret_cast<const half8 *>(L.w_bwd_f16x3);
    const int KT = (L.k_x + L.k_e) >> 5;
    const int frag = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = frag >> 1, part = frag & 1, kg = lane >> 4;
    // kk: hidden input of layer l = output feature here
    const int half = s & 1, s32 = ft == 4 ? s >> 1 : s;
    const int kk = ft == 4 ? 64 * wave + 16 * (2 * half + f) + (lane & 15) : 32 * wave + 16 * f + (lane & 15);
    const int s16 = 2 * s32 + (kg >> 1), t = kk >> 5, lane_src = (kk & 31) + 32 * (kg & 1);
    dst[((size_t)wave * Gw + g0 + g) * 256 + threadIdx.x] = wb[(((size_t)s16 * KT + t) * 2 + part) * 64 + lane_src];
}

// ================================================================================================
// SDF value + last-hidden features + input gradient (normals) of a point list on the fragment stream: get_rbg_value's
// implicit_network(x) + gradient(x) (reference implicit_differentiable_renderer.py:85-104,533-540) for the nets
// vg_shape() takes; the generic 32-row kernel (nefii_mlp.hip) spends ~1 ms per tile on the same work.
// One workgroup of 8 waves per CU, 16 QT rows per tile:
//   forward  - the "16q" evaluator (mlp_tile.h), its epilogue also stashing 16 h_l (fp32, the accumulator layout: one
//              coalesced 16-byte store per lane and feature-tile x query-tile) in the workgroup's own workspace slot;
//   backward - gz_{NH-1} = w_last * sigma'(h_{NH-1}); then per layer l = NH-1 .. 1 the SAME k-loop over the transposed
//              layer's units (contraction over the layer's outputs, gz_l as the LDS image) and an epilogue that
//              multiplies by sigma'(h_{l-1}) from the stash; the encoding columns' gradient of the skip layer and of
//              layer 0 comes from a 32x32x16 side GEMM of waves 0-3 over the layer's own transposed fragments, kept
//              as fp32 in the LDS image's (free by then) encoding columns; last the chain through the encoding.
// ================================================================================================
template <int FT>
__device__ __forceinline__ float *vg_ge_row(LdsQ<FT> &lds, int te, int row) {
    return reinterpret_cast<float *>((te ? lds.Xl : lds.Xh) + row * QGeo<FT>::XP + QGeo<FT>::HW);
}

// GE[row][32 te ..] += gz_l[row][:] . W_l[:, encoding columns 32 te ..] for the layer's two encoding tiles; one wave per
// (encoding tile, 32-row tile): waves 0..3 (64-row tiles) / 0..5 (96-row tiles)
template <int QT, int FT>
__device__ __forceinline__ void vg_enc_grad(const nefii_layer &L, LdsQ<FT> &lds, int wave, int lane, float inv_scale) {
    static_assert(QT % 2 == 0 && QT <= 8, "whole 32-row tiles, one wave each");
    constexpr int XP = QGeo<FT>::XP, EP = QGeo<FT>::HW;
    if (wave >= QT) return;
    const int te = wave & 1, rt = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int KT = (L.k_x + L.k_e) >> 5, t = (L.k_x >> 5) + te, S = L.n_pad >> 4;
    const half8 *wb = reinterpret_cast<const half8 *>(L.w_bwd_f16x3) + ((size_t)t * 2) * 64 + lane;
    const _Float16 *ah = lds.Xh + (32 * rt + r) * XP + 8 * h + (EP - L.n_pad);
    const _Float16 *al = lds.Xl + (32 * rt + r) * XP + 8 * h + (EP - L.n_pad);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int s0 = 0; s0 < S; s0 += 4) {        // S = 32: four 16-deep steps per trip, their fragment loads issued together
        half8 wh[4], wl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wh[u] = wb[(size_t)(s0 + u) * KT * 128], wl[u] = wb[(size_t)(s0 + u) * KT * 128 + 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const half8 xh = *reinterpret_cast<const half8 *>(ah + 16 * (s0 + u));
            const half8 xl = *reinterpret_cast<const half8 *>(al + 16 * (s0 + u));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xl, acc, 0, 0, 0);
        }
    }
    float *ge = vg_ge_row<FT>(lds, te, 32 * rt + r);       // this lane owns its 16 slots of the row: plain read-modify-write
#pragma unroll
    for (int i = 0; i < 16; ++i) ge[(i & 3) + 8 * (i >> 2) + 4 * h] += acc[i] * inv_scale;
}

template <int QT, int FT>
__global__ __launch_bounds__(512, 2) void sdf_value_grad16q_kernel(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                                  float *__restrict__ sdf_out, int out_stride,
                                                                  float *__restrict__ feat_out, int feat_stride,
                                                                  float *__restrict__ grad_out, float4v *__restrict__ ws,
                                                                  size_t vg_off, int Gw) {
    NEFII_CLAIM_SIMD_2();
    constexpr int ROWS = 16 * QT, NW = 8, RT = QT / 2, XP = QGeo<FT>::XP, EP = QGeo<FT>::HW, NJ = FT * QT;
    static_assert(ROWS <= QGeo<FT>::ROWS, "rows of the LDS image");
    __shared__ LdsQ<FT> lds;
    __shared__ float raw[ROWS * 9];
    __shared__ float psum[NW * ROWS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int NH = m.n_layers - 1;
    const float inv_scale = 1.f / (W16_SCALE * A16_SCALE), k16 = inv_scale * A16_SCALE;
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    // stash element (layer l, accumulator j) of this lane: stash[(l * NW * NJ + j) * 64]
    float4v *stash = ws + ((size_t)blockIdx.x * NH * NW + wave) * NJ * 64 + lane;
    constexpr size_t SL = (size_t)NW * NJ * 64;             // float4 per layer
    P16<8>::Stage b[4];
    PCursor cur;
    cur.bytes = (unsigned)Gw * 4096;
    cur.base = reinterpret_cast<const half8 *>(m.w_stream) + vg_off + (size_t)wave * Gw * 256 + lane;
    cur.off = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) pload<8>(b[u], cur);
    const int boff = 16 * FT * wave + (lane & (16 * FT - 1));
    const _Float16 *qh0 = lds.Xh + (lane & 15) * XP + 8 * (lane >> 4), *ql0 = lds.Xl + (lane & 15) * XP + 8 * (lane >> 4);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * ROWS;
        for (int i = tid; i < ROWS * 9; i += 512) {
            const int p = i / 9, c = i - 9 * p;
            int64_t idx = base + p;
            if (idx >= n) idx = n - 1;
            raw[i] = c < 3 ? x[idx * 3 + c] : 0.f;
        }
        float bnext = m.layer[0].bias[boff];
        __syncthreads();
        encode_tile16q<FT>(m, raw, lds, ROWS);
        __syncthreads();
        // ------------------------------------------------ forward
        for (int l = 0; l < NH; ++l) {
            const nefii_layer &L = m.layer[l];
            const int units = q_units<FT>(L);
            const _Float16 *ah = qh0 + (EP - L.k_x), *al = ql0 + (EP - L.k_x);
            const float *bp = m.layer[l + 1 < NH ? l + 1 : l].bias + boff;
            asm volatile("" ::"s"(units), "v"(ah), "v"(al), "v"(bp));
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_sched_barrier(0);
            const float bvec = bnext;
            f32x4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
            QAct<QT> a[2];
            qload_a<QT, XP>(a[0], ah, al, 0);
            qgemm<QT, FT>(units, b, a, cur, ah, al, acc);
            bnext = *bp;
            __builtin_amdgcn_sched_barrier(0);
            half4 phi[NJ], plo[NJ];
            const int bsrc = __builtin_bit_cast(int, bvec * A16_SCALE);
            float4v *st = stash + (size_t)l * SL;
            const bool feat = feat_out != nullptr && l == NH - 1;
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                float4v bs;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * ft + 4 * (lane >> 4) + k), bsrc));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int j = ft * QT + qt;
                    float4v hs;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hs[k] = softplus100_s16(__builtin_fmaf(acc[j][k], k16, bs[k]));
                    st[j * 64] = hs;
                    const half4 hi = __builtin_convertvector(hs, half4);
                    phi[j] = hi;
                    plo[j] = __builtin_convertvector(hs - __builtin_convertvector(hi, float4v), half4);
                    if (feat) {
                        const int64_t row = base + 16 * qt + (lane & 15);
                        const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
                        if (row < n) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (f0 + k < L.n_out) feat_out[(size_t)row * feat_stride + f0 + k] = hs[k] * (1.f / A16_SCALE);
                        }
                    }
                }
            }
            __syncthreads();
            _Float16 *xh = lds.Xh + (EP - L.n_pad), *xl = lds.Xl + (EP - L.n_pad);
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                    *reinterpret_cast<half4 *>(xl + query * XP + f0) = plo[ft * QT + qt];
                }
            }
            __syncthreads();
        }
        // last layer: 32x32x16 fragments of the layer's own w_f16x3.  One column (use_last_as_f nets): K split over the
        // waves (as in "16q"); a wide one (SDF + feature columns): whole (column tile, row tile) products per wave.
        if (m.layer[NH].n_pad > 32) {
            const int r = lane & 31, h = lane >> 5;
            const nefii_layer &L = m.layer[NH];
            const int NT = L.n_pad >> 5, S = L.k_x >> 4;
            const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
            for (int c = wave; c < NT * RT; c += NW) {
                const int t = c / RT, rt = c - t * RT;
                const _Float16 *ah = lds.Xh + (32 * rt + r) * XP + 8 * h + (EP - L.k_x);
                const _Float16 *al = lds.Xl + (32 * rt + r) * XP + 8 * h + (EP - L.k_x);
                f32x16 acc2;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
                for (int s0 = 0; s0 < S; s0 += 4) {
                    half8 wh[4], wlo[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        wh[u] = wl[((size_t)(s0 + u) * NT + t) * 128], wlo[u] = wl[((size_t)(s0 + u) * NT + t) * 128 + 64];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const half8 xh8 = *reinterpret_cast<const half8 *>(ah + 16 * (s0 + u));
                        const half8 xl8 = *reinterpret_cast<const half8 *>(al + 16 * (s0 + u));
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xh8, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[u], xh8, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xl8, acc2, 0, 0, 0);
                    }
                }
                const int64_t row = base + 32 * rt + r;
                if (row < n) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int f = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (f < L.n_out) sdf_out[(size_t)row * out_stride + f] = acc2[i] * inv_scale + L.bias[f];
                    }
                }
            }
            __syncthreads();        // every wave is done reading h_{NH-1} from the image
        } else {
            const int r = lane & 31, h = lane >> 5;
            const nefii_layer &L = m.layer[NH];
            const int NT = L.n_pad >> 5;
            const half8 *wl = reinterpret_cast<const half8 *>(L.w_f16x3) + lane;
            const _Float16 *ah = lds.Xh + r * XP + 8 * h + (EP - L.k_x), *al = lds.Xl + r * XP + 8 * h + (EP - L.k_x);
            const int ksw = (L.k_x >> 4) / NW;
            f32x16 acc2[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[rt][i] = 0.f;
            for (int u = 0; u < ksw; ++u) {
                const int s = wave * ksw + u;
                const half8 wh = wl[(size_t)s * NT * 128], wlo = wl[(size_t)s * NT * 128 + 64];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const half8 xh8 = *reinterpret_cast<const half8 *>(ah + rt * 32 * XP + 16 * s);
                    const half8 xl8 = *reinterpret_cast<const half8 *>(al + rt * 32 * XP + 16 * s);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh8, acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh8, acc2[rt], 0, 0, 0);
                    acc2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl8, acc2[rt], 0, 0, 0);
                }
            }
            if (h == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) psum[wave * ROWS + 32 * rt + r] = acc2[rt][0];
            }
            __syncthreads();        // partial sums complete; every wave is done reading h_{NH-1} from the image
            if (tid < ROWS) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += psum[w * ROWS + tid];
                if (base + tid < n) sdf_out[(size_t)(base + tid) * out_stride] = sum * inv_scale + L.bias[0];
            }
        }
        // ------------------------------------------------ backward
        // seed: gz_{NH-1}[row][f] = W_last[0][f] * sigma'(h_{NH-1}[row][f]); the encoding columns become the fp32 GE rows
        {
            const nefii_layer &LL = m.layer[NH];
            const half8 *wlb = reinterpret_cast<const half8 *>(LL.w_bwd_f16x3);       // s16 = 0: outputs 0..7 of lanes 0..31
            const float4v *st = stash + (size_t)(NH - 1) * SL;
            for (int i = tid; i < ROWS * 32; i += 512) {
                vg_ge_row<FT>(lds, 0, i >> 5)[i & 31] = 0.f;
                vg_ge_row<FT>(lds, 1, i >> 5)[i & 31] = 0.f;
            }
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
                float4v wv;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = f0 + k;
                    const half8 hi = wlb[((size_t)(f >> 5) * 2) * 64 + (f & 31)], lo = wlb[((size_t)(f >> 5) * 2 + 1) * 64 + (f & 31)];
                    wv[k] = ((float)hi[0] + (float)lo[0]) * (1.f / W16_SCALE);
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float4v hs = st[(ft * QT + qt) * 64];
                    const int query = 16 * qt + (lane & 15);
                    float4v v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = wv[k] * softplus100_bwd_fast(hs[k] * (1.f / A16_SCALE)) * A16_SCALE;
                    const half4 hi = __builtin_convertvector(v, half4);
                    *reinterpret_cast<half4 *>(lds.Xh + query * XP + f0) = hi;
                    *reinterpret_cast<half4 *>(lds.Xl + query * XP + f0) =
                        __builtin_convertvector(v - __builtin_convertvector(hi, float4v), half4);
                }
            }
            __syncthreads();
        }
        for (int l = NH - 1; l >= 1; --l) {
            const nefii_layer &L = m.layer[l];
            if (L.k_e) vg_enc_grad<QT, FT>(L, lds, wave, lane, inv_scale);
            const int units = vg_units_layer(L.n_pad, FT);
            const _Float16 *ah = qh0 + (EP - L.n_pad), *al = ql0 + (EP - L.n_pad);
            asm volatile("" ::"s"(units), "v"(ah), "v"(al));
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
            QAct<QT> a[2];
            qload_a<QT, XP>(a[0], ah, al, 0);
            qgemm<QT, FT>(units, b, a, cur, ah, al, acc);
            half4 phi[NJ], plo[NJ];
            const float4v *st = stash + (size_t)(l - 1) * SL;
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                float4v hs[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) hs[qt] = st[(ft * QT + qt) * 64];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int j = ft * QT + qt;
                    float4v v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[j][k] * k16 * softplus100_bwd_fast(hs[qt][k] * (1.f / A16_SCALE));
                    const half4 hi = __builtin_convertvector(v, half4);
                    phi[j] = hi;
                    plo[j] = __builtin_convertvector(v - __builtin_convertvector(hi, float4v), half4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            _Float16 *xh = lds.Xh + (EP - L.k_x), *xl = lds.Xl + (EP - L.k_x);
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                const int f0 = 16 * FT * wave + 16 * ft + 4 * (lane >> 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int query = 16 * qt + (lane & 15);
                    *reinterpret_cast<half4 *>(xh + query * XP + f0) = phi[ft * QT + qt];
                    *reinterpret_cast<half4 *>(xl + query * XP + f0) = plo[ft * QT + qt];
                }
            }
            __syncthreads();
        }
        vg_enc_grad<QT, FT>(m.layer[0], lds, wave, lane, inv_scale);
        __syncthreads();
        // chain through the positional encoding
        if (tid < ROWS * 3) {
            const int p = tid / 3, c = tid - 3 * p;
            if (base + p < n) {
                const float *v = raw + p * 9;
                const int w0 = enc_width(m.enc_freqs[0]);
                const float *g0 = vg_ge_row<FT>(lds, 0, p), *g1 = vg_ge_row<FT>(lds, 1, p);
                float g = 0.f;
                for (int col = 0; col < w0; ++col) {
                    int comp;
                    const float d = enc_deriv(v, col, comp);
                    if (comp == c) g += (col < 32 ? g0[col] : g1[col - 32]) * d;
                }
                grad_out[(size_t)(base + p) * 3 + c] = g;
            }
        }
        __syncthreads();
    }
}

// the same tile evaluator over an explicit point list (nefii_sdf_eval)
__global__ __launch_bounds__(512, 2) void sdf_points_kernel16w(nefii_mlp m, const float *__restrict__ x, int64_t n,
                                                              float *__restrict__ out) {
    NEFII_CLAIM_SIMD_2();
    __shared__ Lds16w lds;
    __shared__ float raw[TILE_W * 9];
    __shared__ float *dest[TILE_W];
    const int64_t n_tiles = (n + TILE_W - 1) / TILE_W;
    int ke = 0;
    for (int l = 0; l < m.n_layers; ++l) ke = m.layer[l].k_e > ke ? m.layer[l].k_e : ke;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tid = threadIdx.x;
        if (tid < TILE_W) {
            const int64_t q = tile * TILE_W + tid;
            float *rw = raw + tid * 9;
            const bool live = q < n;
            rw[0] = live ? x[q * 3] : 0.f, rw[1] = live ? x[q * 3 + 1] : 0.f, rw[2] = live ? x[q * 3 + 2] : 0.f;
            rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = rw[8] = 0.f;
            dest[tid] = live ? out + q : nullptr;
        }
        __syncthreads();
        sdf_tile16w(m, lds, raw, dest, ke);
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// rows of min-SDF draws of a call (nefii_tracer_params.minsdf_group)
int64_t minsdf_rows(int64_t n, const nefii_tracer_params *p) {
    return p->minsdf_group > 0 ? (n + p->minsdf_group - 1) / p->minsdf_group : 1;
}

size_t carve(RayState &s, char *base, int64_t n, int ns, int cap, int64_t step_rows) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    float **fl[] = {&s.t_s, &s.t_e, &s.cur_s, &s.cur_e, &s.nxt_s, &s.nxt_e, &s.t_min,
                    &s.t_max, &s.res_s, &s.res_e, &s.lo, &s.hi, &s.mid};
    for (auto f : fl) *f = (float *)take(sizeof(float) * n);
    s.flags = (int *)take(sizeof(int) * n);
    s.big = (float *)take(sizeof(float) * (size_t)n * ns);
    s.singles = (unsigned *)take(sizeof(unsigned) * 2 * n);
    s.dense = (unsigned *)take(sizeof(unsigned) * n);
    s.tri = (unsigned *)take(sizeof(unsigned) * n);
    s.cdense = (unsigned *)take(sizeof(unsigned) * 4 * n);
    s.refine = (unsigned *)take(sizeof(unsigned) * (size_t)n * (cap > 0 ? cap : 0));
    s.csingles = (unsigned *)take(sizeof(unsigned) * 2 * n);
    s.block_live = (int *)take(sizeof(int) * ((n + 255) / 256));
    s.crefine = (unsigned *)take(step_rows > 0 ? sizeof(unsigned) * (size_t)n * CREF_CAP : 0);
    s.ord = (unsigned char *)take(step_rows > 0 ? (size_t)step_rows * ns : 0);
    return off;
}

}  // namespace

// ---- optional per-launch timing (HIP events on the launch stream; off by default) ------------------
namespace {
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs: [2i] before, [2i+1] after each eval launch
    size_t used = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // whole nefii_trace_rays call
    bool have_span = false;
} g_prof;
hipEvent_t prof_event() {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof.ev.push_back(e);
    }
    return g_prof.ev[g_prof.used++];
}
}  // namespace

extern "C" int nefii_trace_profile_enable(int on) {
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.have_span = false;
    return 0;
}

// Sum of the eval-kernel launch durations recorded since enable (ms), their count, and the span of the
// last nefii_trace_rays call (ms).  Synchronises on the recorded events.
extern "C" int nefii_trace_profile_read(double *eval_ms, int *n_eval, double *span_ms) {
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        hipError_t e = hipEventSynchronize(g_prof.ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
    }
    if (eval_ms) *eval_ms = tot;
    if (n_eval) *n_eval = (int)(g_prof.used / 2);
    if (span_ms) {
        *span_ms = 0.0;
        if (g_prof.have_span) {
            (void)hipEventSynchronize(g_prof.t1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, g_prof.t0, g_prof.t1);
            *span_ms = ms;
        }
    }
    g_prof.used = 0;
    return 0;
}

// Per-launch durations (ms) of the recorded eval launches, in launch order; does not clear the record.
extern "C" int nefii_trace_profile_launches(float *h_ms, int cap) {
    int n = 0;
    for (size_t i = 0; i + 1 < g_prof.used && n < cap; i += 2, ++n) {
        hipError_t e = hipEventSynchronize(g_prof.ev[i + 1]);
        if (e != hipSuccess) return -(int)e;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]);
        h_ms[n] = ms;
    }
    return n;
}

// bytes of the first four copies (the fifth - the "16f" evaluator's - follows them)
static size_t stream_bytes_1to4(const nefii_mlp *h_sdf) {
    size_t units_vg = 0;
    if (vg_shape(h_sdf)) units_vg = (size_t)stream_steps(h_sdf) + vg_units_bwd(h_sdf);
    return (size_t)8 * (stream_steps(h_sdf) + stream_steps8(h_sdf) + units_vg) * 256 * sizeof(half8) +
           (size_t)8 * stream_steps_sp(h_sdf) * shape16p(h_sdf) * 64 * sizeof(half8);
}
// units per wave of the fifth copy (mlp_tile.h "16f": 8 per 128-deep chunk), 0 when the net does not take that evaluator
int stream_steps_f8(const nefii_mlp *m) {
    if (shape16p(m) != 4 || m->reserved != 1) return 0;
    int G = 0;
    for (int l = 0; l < m->n_layers - 1; ++l) G += 8 * f_chunks(m->layer[l]);
    return G;
}

extern "C" size_t nefii_sdf_stream_bytes(const nefii_mlp *h_sdf) {
    if (!h_sdf || h_sdf->n_layers < 2 || h_sdf->n_layers > NEFII_MAX_LAYERS || !shape16p(h_sdf)) return 0;
    return stream_bytes_1to4(h_sdf) + (size_t)8 * stream_steps_f8(h_sdf) * 4096;
}

extern "C" int nefii_sdf_fp8corr_supported(const nefii_mlp *h_sdf) {
    return h_sdf && h_sdf->w_stream && h_sdf->n_layers >= 2 && h_sdf->n_layers <= NEFII_MAX_LAYERS && stream_steps_f8(h_sdf) > 0;
}

// fifth copy.  Unit g of a wave: chunk c = (g - first unit of its layer) >> 3, j = & 7.  j < 4: the hi fragments of k-step 4 c + j
// (as the single-pass copy).  j >= 4: feature tile ft = j - 4, 16-byte element i of the lane: part = i >> 1 (0: w_l, 1: w_h),
// bytes 16 (i & 1) .. + 15 of the lane's 32: e4m3(W_part[n = 64 wave + 16 ft + (lane & 15)][k = 128 c + 32 (lane >> 4) + byte] x 2^E).
__global__ void pack_sdf_stream_f8_kernel(nefii_mlp m, half8 *__restrict__ dst, int G) {
    const int g = blockIdx.x, wave = blockIdx.y;
    int l = 0, u = g;
    while (u >= 8 * f_chunks(m.layer[l])) u -= 8 * f_chunks(m.layer[l]), ++l;
    const half8 *w = reinterpret_cast<const half8 *>(m.layer[l].w_f16x3);
    const int NT = m.layer[l].n_pad >> 5, K16 = (m.layer[l].k_x + m.layer[l].k_e) >> 4;
    const int c = u >> 3, j = u & 7;
    const int i = threadIdx.x >> 6, lane = threadIdx.x & 63, kg = lane >> 4;
    half8 v;
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
    if (j < 4) {
        const int s = 4 * c + j, n = 64 * wave + 16 * i + (lane & 15);
        const int s16 = 2 * s + (kg >> 1), t = n >> 5, lane_src = (n & 31) + 32 * (kg & 1);
        if (s16 < K16) v = w[(((size_t)s16 * NT + t) * 2) * 64 + lane_src];
    } else {
        const int ft = j - 4, part = i >> 1, n = 64 * wave + 16 * ft + (lane & 15), t = n >> 5;
        const int k0 = 128 * c + 32 * kg + 16 * (i & 1), s16 = k0 >> 4;
        const float sc = f8_scale(part == 0 ? F8_WL_E : F8_WH_E);
        int out[4] = {0, 0, 0, 0};
        if (s16 < K16) {
            const half8 a = w[(((size_t)s16 * NT + t) * 2 + (part == 0 ? 1 : 0)) * 64 + (n & 31)];            // k0 .. k0 + 7
            const half8 b = w[(((size_t)s16 * NT + t) * 2 + (part == 0 ? 1 : 0)) * 64 + (n & 31) + 32];       // k0 + 8 .. + 15
            out[0] = f8_pack4((float)a[0] * sc, (float)a[1] * sc, (float)a[2] * sc, (float)a[3] * sc);
            out[1] = f8_pack4((float)a[4] * sc, (float)a[5] * sc, (float)a[6] * sc, (float)a[7] * sc);
            out[2] = f8_pack4((float)b[0] * sc, (float)b[1] * sc, (float)b[2] * sc, (float)b[3] * sc);
            out[3] = f8_pack4((float)b[4] * sc, (float)b[5] * sc, (float)b[6] * sc, (float)b[7] * sc);
        }
        const i32x4 o4 = {out[0], out[1], out[2], out[3]};
        v = __builtin_bit_cast(half8, o4);
    }
    dst[((size_t)wave * G + g) * 256 + threadIdx.x] = v;
}

extern "C" int nefii_sdf_coarse_supported(const nefii_mlp *h_sdf) {
    return h_sdf && h_sdf->w_stream && h_sdf->n_layers >= 2 && h_sdf->n_layers <= NEFII_MAX_LAYERS &&
           stream_steps_sp(h_sdf) > 0;
}

extern "C" int nefii_pack_sdf_stream(const nefii_mlp *h_sdf, void *w_stream, void *stream) {
    if (!h_sdf || !w_stream) return NEFII_E_ARG;
    if (nefii_sdf_stream_bytes(h_sdf) == 0) return NEFII_E_UNSUPPORTED;
    for (int l = 0; l < h_sdf->n_layers - 1; ++l)
        if (!h_sdf->layer[l].w_f16x3) return NEFII_E_ARG;
    const int G = stream_steps(h_sdf);
    hipLaunchKernelGGL(pack_sdf_stream_kernel, dim3(G, 8), dim3(256), 0, (hipStream_t)stream, *h_sdf, (half8 *)w_stream, G,
                       shape16p(h_sdf), 0);
    HIP_CHECK_LAUNCH();
    const int G8 = stream_steps8(h_sdf);
    if (G8 > 0) {
        hipLaunchKernelGGL(pack_sdf_stream_kernel, dim3(G8, 8), dim3(256), 0, (hipStream_t)stream, *h_sdf,
                           (half8 *)w_stream + (size_t)8 * G * 256, G8, 4, 1);
        HIP_CHECK_LAUNCH();
    }
    const int Gs = stream_steps_sp(h_sdf);
    if (Gs > 0) {
        const int ft = shape16p(h_sdf);
        hipLaunchKernelGGL(pack_sdf_stream_sp_kernel, dim3(Gs, 8), dim3(64 * ft), 0, (hipStream_t)stream, *h_sdf,
                           (half8 *)w_stream + (size_t)8 * (G + G8) * 256, Gs, ft);
        HIP_CHECK_LAUNCH();
    }
    if (const int ft = vg_shape(h_sdf)) {
        const int Gb = vg_units_bwd(h_sdf), Gw = G + Gb;
        half8 *dst = (half8 *)w_stream + vg_stream_offset(h_sdf);
        hipLaunchKernelGGL(pack_sdf_stream_kernel, dim3(G, 8), dim3(256), 0, (hipStream_t)stream, *h_sdf, dst, Gw, ft, 0);
        HIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(pack_sdf_stream_bwd_kernel, dim3(Gb, 8), dim3(256), 0, (hipStream_t)stream, *h_sdf, dst, Gw, G, ft);
        HIP_CHECK_LAUNCH();
    }
    if (const int Gf = stream_steps_f8(h_sdf)) {
        hipLaunchKernelGGL(pack_sdf_stream_f8_kernel, dim3(Gf, 8), dim3(256), 0, (hipStream_t)stream, *h_sdf,
                           reinterpret_cast<half8 *>((char *)w_stream + stream_bytes_1to4(h_sdf)), Gf);
        HIP_CHECK_LAUNCH();
    }
    return 0;
}

namespace nefii {
static int vg_grid(int64_t n, int rows) {
    const int64_t n_tiles = (n + rows - 1) / rows;
    return (int)(n_tiles < 256 ? (n_tiles > 0 ? n_tiles : 1) : 256);     // one workgroup per CU, grid-strided
}
// NEFII_VG_STREAM=0: keep the generic 32-row kernel (A/B measurements)
static bool vg_enabled() {
    static const bool v = [] {
        const char *e = getenv("NEFII_VG_STREAM");
        return !(e && atoi(e) == 0);
    }();
    return v;
}
size_t value_grad_stream_ws_bytes(const nefii_mlp *m, int64_t n) {
    if (!m || n <= 0 || !m->w_stream || !vg_enabled()) return 0;
    const int ft = vg_shape(m);
    if (!ft) return 0;
    const int rows = ft == 4 ? 64 : 96;        // stash: one float4 per lane and (feature tile, query tile) of a wave
    return (size_t)vg_grid(n, rows) * (m->n_layers - 1) * 8 * (ft * rows / 16) * 64 * sizeof(float4v);
}
int value_grad_stream_launch(const nefii_mlp *m, const float *x, int64_t n, float *sdf_out, int out_stride, float *feat_out,
                             int feat_stride, float *grad_out, float *ws, hipStream_t st) {
    const int ft = m->w_stream && vg_enabled() ? vg_shape(m) : 0;
    if (!ft) return NEFII_E_UNSUPPORTED;
    const int Gw = stream_steps(m) + vg_units_bwd(m);
    if (ft == 4)
        hipLaunchKernelGGL((sdf_value_grad16q_kernel<4, 4>), dim3(vg_grid(n, 64)), dim3(512), 0, st, *m, x, n, sdf_out, out_stride,
                           feat_out, feat_stride, grad_out, reinterpret_cast<float4v *>(ws), vg_stream_offset(m), Gw);
    else
        hipLaunchKernelGGL((sdf_value_grad16q_kernel<6, 2>), dim3(vg_grid(n, 96)), dim3(512), 0, st, *m, x, n, sdf_out, out_stride,
                           feat_out, feat_stride, grad_out, reinterpret_cast<float4v *>(ws), vg_stream_offset(m), Gw);
    HIP_CHECK_LAUNCH();
    return 0;
}
}  // namespace nefii

// NEFII_COARSE_D=1: 512-wide nets' 64-query single-pass tiles on the two-group form ("16d", mlp_tile.h; bit-identical values).
// Round 5, same box: 59.6 -> 58.2 us per tile on full rounds, 71.2 -> 84.6 at one tile per CU; config 3 184.1 -> 183.9 ms per
// step, config 2 2.72 -> 2.75, config 1 0.815 -> 0.849: the tile gains in cycles (layer period 16.0 k -> 11.4 k per tile) what
// the chip takes back in clock (profiles/r05/two_group_tile/).  As two 4-wave workgroups per CU (no register claim: the
// packed-fp32 canary fails beside it) config 3 went 185.6 -> 178.7.  Default: the eight-wave form "16s".
// workgroups of the two-group form: group 0 takes tiles [0, grid), group 1 [grid, 2 grid): one per CU keeps both groups of every
// CU busy from 257 tiles on
static unsigned coarse_d_grid() {
    static const unsigned v = [] {
        const char *e = getenv("NEFII_COARSE_D_GRID");
        const int g = e ? atoi(e) : 0;
        return (unsigned)(g > 0 ? g : 256);
    }();
    return v;
}
static int coarse_two_groups() {
    static const int v = [] {
        const char *e = getenv("NEFII_COARSE_D");
        return e ? atoi(e) : 0;
    }();
    return v;
}
// queries per tile of the single-pass evaluator, 16 * QT.  512-wide nets: QT 4 (default) / 6 / 8 (NEFII_COARSE_QT; the big
// tiles read activation fragments single-buffered and run their epilogue behind the barrier to fit 256 registers).
// Measured per 64 queries at 12 tiles per CU: 62.0 / 58.9 / 58.3 us, at one tile per CU 77 / 100 / 123 us - the tile is
// bound by its matrix work plus the epilogue's VALU work, not by the fragment stream, so bigger tiles buy ~5 % on
// full rounds and cost latency on small ones
static int coarse_qt() {
    static const int v = [] {
        const char *e = getenv("NEFII_COARSE_QT");
        const int q = e ? atoi(e) : 0;
        return q == 4 || q == 6 || q == 8 ? q : 0;
    }();
    return v;
}
// rows of a coarse tile for a net of feature-tile count ft
static int coarse_rows(int ft) {
    const int q = coarse_qt();
    if (ft == 2) return 96;
    return 16 * (q ? q : 4);
}
// launches KERNEL<QT, FT, DB> for the configured tile
#define NEFII_COARSE_LAUNCH(KERNEL, KERNEL_D, ft, grid, st, ...)                                                    \
    do {                                                                                                            \
        const int rows_ = coarse_rows(ft);                                                                          \
        if ((ft) == 2)                                                                                              \
            hipLaunchKernelGGL((KERNEL<6, 2>), grid, dim3(512), 0, st, __VA_ARGS__);                                \
        else if (rows_ == 64 && coarse_two_groups())                                                                \
            hipLaunchKernelGGL((KERNEL_D<4>), dim3((grid).x < coarse_d_grid() ? (grid).x : coarse_d_grid()), dim3(512), 0, st, \
                               __VA_ARGS__);                                                                    \
        else if (rows_ == 64)                                                                                       \
            hipLaunchKernelGGL((KERNEL<4, 4>), grid, dim3(512), 0, st, __VA_ARGS__);                                \
        else if (rows_ == 96)                                                                                       \
            hipLaunchKernelGGL((KERNEL<6, 4, false>), grid, dim3(512), 0, st, __VA_ARGS__);                         \
        else                                                                                                        \
            hipLaunchKernelGGL((KERNEL<8, 4, false>), grid, dim3(512), 0, st, __VA_ARGS__);                         \
    } while (0)

extern "C" int nefii_sdf_eval_coarse(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream) {
    if (!h_sdf || h_sdf->n_layers < 1 || h_sdf->n_layers > NEFII_MAX_LAYERS) return NEFII_E_ARG;
    if (!nefii_sdf_coarse_supported(h_sdf)) return NEFII_E_UNSUPPORTED;
    if (n <= 0) return 0;
    if (!x || !sdf_out) return NEFII_E_ARG;
    if (h_sdf->enc_freqs[0] < 0 || h_sdf->enc_freqs[1] >= 0 || h_sdf->enc_freqs[2] >= 0 || h_sdf->feat_width != 0 ||
        h_sdf->layer[0].k_x != 0)
        return NEFII_E_UNSUPPORTED;
    const int ft = shape16p(h_sdf);
    const int rows = coarse_rows(ft);
    const int64_t n_tiles = (n + rows - 1) / rows;
    const dim3 grid((int)(n_tiles < 512 ? n_tiles : 512));
    NEFII_COARSE_LAUNCH(sdf_points_kernel16s, sdf_points_kernel16d, ft, grid, (hipStream_t)stream, *h_sdf, x, n, sdf_out);
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_sdf_eval_fp8corr(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream) {
    if (!h_sdf || !x || !sdf_out || n < 0) return NEFII_E_ARG;
    if (!nefii_sdf_fp8corr_supported(h_sdf)) return NEFII_E_UNSUPPORTED;
    for (int l = 0; l < h_sdf->n_layers; ++l)
        if (!h_sdf->layer[l].w_f16x3) return NEFII_E_ARG;
    if (n == 0) return 0;
    const int64_t n_tiles = (n + 63) / 64;
    hipLaunchKernelGGL(sdf_points_kernel16f, dim3((int)(n_tiles < 512 ? n_tiles : 512)), dim3(512), 0, (hipStream_t)stream, *h_sdf, x,
                       n, sdf_out, (const void *)((const char *)h_sdf->w_stream + stream_bytes_1to4(h_sdf)));
    HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int nefii_sdf_eval(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream) {
    if (!h_sdf || h_sdf->n_layers < 1 || h_sdf->n_layers > NEFII_MAX_LAYERS) return NEFII_E_ARG;
    if (n <= 0) return 0;
    if (!x || !sdf_out) return NEFII_E_ARG;
    if (h_sdf->enc_freqs[0] < 0 || h_sdf->enc_freqs[1] >= 0 || h_sdf->enc_freqs[2] >= 0 || h_sdf->feat_width != 0 ||
        h_sdf->layer[0].k_x != 0)
        return NEFII_E_UNSUPPORTED;
    for (int l = 0; l < h_sdf->n_layers; ++l)
        if (!h_sdf->layer[l].w_f16x3 || !h_sdf->layer[l].bias) return NEFII_E_ARG;
    const int64_t n_tiles = (n + TILE_W - 1) / TILE_W;
    const int ft = fits16p(h_sdf);
    if (ft == 2)
        hipLaunchKernelGGL(sdf_points_kernel16q<2>, dim3((int)(n_tiles < 512 ? n_tiles : 512)), dim3(512), 0,
                           (hipStream_t)stream, *h_sdf, x, n, sdf_out);
    else if (ft && h_sdf->reserved == 1)
        hipLaunchKernelGGL(sdf_points_kernel16q<4>, dim3((int)(n_tiles < 512 ? n_tiles : 512)), dim3(512), 0,
                           (hipStream_t)stream, *h_sdf, x, n, sdf_out);
    else if (ft)
        hipLaunchKernelGGL(sdf_points_kernel16p<P16W>, dim3((int)(n_tiles < 512 ? n_tiles : 512)), dim3(64 * P16W), 0,
                           (hipStream_t)stream, *h_sdf, x, n, sdf_out);
    else
        hipLaunchKernelGGL(sdf_points_kernel16w, dim3((int)(n_tiles < 512 ? n_tiles : 512)), dim3(WG_W), 0,
                           (hipStream_t)stream, *h_sdf, x, n, sdf_out);
    HIP_CHECK_LAUNCH();
    return 0;
}

// samples of one ray the coarse pass refines individually (0: coarse pass off)
static int coarse_cap(const nefii_tracer_params *p) {
    if (!(p->coarse_tau > 0.f)) return 0;
    // break-even of refining k samples of a ray (k split evaluations on top of the coarse pass, 1/3 each) against
    // re-evaluating all 100 in split precision is k ~ 67
    const int c = p->coarse_cap <= 0 ? 64 : p->coarse_cap;
    return c > 100 ? 100 : c;
}

// the staged min-SDF search runs (the coarse pass itself may still be refused for the net: then it is simply not used)
static bool minsdf_staged(const nefii_tracer_params *p) {
    return p->coarse_tau > 0.f && p->minsdf_lipschitz > 0.f && p->n_steps >= 16 && p->n_steps <= 128;
}

extern "C" int nefii_trace_max_rounds(const nefii_tracer_params *p) {
    if (!p) return 0;
    // initial eval + iters*(step + back-offs) -> sampler -> bisection (L levels per round) -> min-SDF -> bookkeeping
    // with the coarse pass each of the two dense searches takes one round more (coarse samples -> refined samples)
    const int L = p->bisect_levels >= 1 && p->bisect_levels <= 5 ? p->bisect_levels : 3;
    // ... and the bracket search one more for the rays whose leading samples (evaluated exactly first) hold no negative one,
    // up to three more for the quarter rows of its coarse pass, the min-SDF search one more for its two-stage refinement
    // tiered sphere tracing: every sphere-tracing evaluation may take a second round (the coarse value, then the exact one)
    const int trace = 1 + p->sphere_tracing_iters * (1 + p->line_step_iters);
    // staged min-SDF search: its two stages take the place of the one coarse round
    return trace + 1 + (p->n_rootfind_steps + L - 1) / L + 1 + 2 +
           (p->coarse_tau > 0.f ? 3 + 3 + 1 + (p->trace_tier ? trace : 0) : 0) + (minsdf_staged(p) ? 1 : 0);
}

extern "C" size_t nefii_trace_workspace_bytes(int64_t n_rays, const nefii_tracer_params *p) {
    if (!p || n_rays <= 0) return 0;
    RayState s;
    size_t bytes = carve(s, nullptr, n_rays, p->n_steps, coarse_cap(p), minsdf_staged(p) ? minsdf_rows(n_rays, p) : 0);
    bytes += align256(sizeof(int) * NCNT * (size_t)nefii_trace_max_rounds(p));
    return bytes;
}

// One tracer invocation (a ray batch on a stream), split into prepare / one round / finish so that several batches can be
// enqueued round by round on separate streams (nefii_trace_rays_groups).
struct TraceJob {
    Params P;
    const nefii_mlp *sdf;
    int precision, rounds, adv_blocks, eval_blocks, eval_blocks_w;
    int pipelined;      // feature tiles per wave of the pipelined evaluator (4 / 2), 0: generic kernels
    int coarse;         // the coarse pass runs (its kernel is launched every round)
    const void *f8;     // nefii_tracer_params.split_fp8 and the net has the fifth stream copy: the "16f" evaluator's stream, else null
    hipStream_t st;
    int32_t *counters;
};

int prepare_job(TraceJob &J, const nefii_mlp *h_sdf, const nefii_tracer_params *h_params, const float *origins,
                const float *dirs, const uint8_t *object_mask, int64_t n_rays, const float *lin_steps,
                const float *minsdf_steps, float *out_points, uint8_t *out_hit, float *out_dists, void *workspace,
                size_t workspace_bytes, int32_t *counters, bool reset, void *stream) {
    if (!h_sdf || !h_params || !origins || !dirs || !object_mask || !lin_steps || !out_points || !out_hit ||
        !out_dists || !workspace)
        return NEFII_E_ARG;
    if (n_rays <= 0 || n_rays >= (1ll << 29)) return NEFII_E_SHAPE;
    if (h_params->training && !minsdf_steps) return NEFII_E_ARG;
    const int levels = h_params->bisect_levels >= 1 && h_params->bisect_levels <= 5 ? h_params->bisect_levels : 3;
    if (h_params->bisect_levels < 0 || h_params->bisect_levels > 5) return NEFII_E_ARG;
    if (h_params->n_steps < (1 << levels) || h_params->sphere_tracing_iters > 250 || h_params->line_step_iters > 15 ||
        h_params->n_rootfind_steps > 250)
        return NEFII_E_SHAPE;
    if (h_sdf->enc_freqs[0] < 0 || h_sdf->enc_freqs[1] >= 0 || h_sdf->enc_freqs[2] >= 0 || h_sdf->feat_width != 0 ||
        h_sdf->layer[0].k_x != 0)
        return NEFII_E_UNSUPPORTED;
    if (workspace_bytes < nefii_trace_workspace_bytes(n_rays, h_params)) return NEFII_E_SHAPE;
    if (h_params->precision < 0 || h_params->precision > 2) return NEFII_E_ARG;
    if (h_params->precision >= 1)
        for (int l = 0; l < h_sdf->n_layers; ++l)
            if (!h_sdf->layer[l].w_f16x3) return NEFII_E_ARG;
    J.sdf = h_sdf;
    J.st = (hipStream_t)stream;
    J.precision = h_params->precision;
    J.rounds = nefii_trace_max_rounds(h_params);
    J.counters = counters;
    Params &P = J.P;
    P.p = *h_params;
    P.n = n_rays;
    P.o = origins;
    P.d = dirs;
    P.obj = object_mask;
    P.lin = lin_steps;
    P.steps = minsdf_steps ? minsdf_steps : lin_steps;
    if (!minsdf_steps) P.p.minsdf_group = 0;
    P.out_pts = out_points;
    P.out_dist = out_dists;
    P.out_hit = out_hit;
    // coarse pass: needs the single-pass stream (pipelined shapes, 16x16x32 layout), sample ids in 7 bits and ray ids in
    // the other 25 of a refine entry
    J.pipelined = h_params->precision == 2 ? fits16p(h_sdf) : 0;
    if (h_params->split_fp8 < 0 || h_params->split_fp8 > 1) return NEFII_E_ARG;
    J.f8 = (h_params->split_fp8 && J.pipelined == 4 && nefii_sdf_fp8corr_supported(h_sdf))
               ? (const void *)((const char *)h_sdf->w_stream + stream_bytes_1to4(h_sdf)) : nullptr;
    J.coarse = h_params->coarse_tau > 0.f && J.pipelined && nefii_sdf_coarse_supported(h_sdf) &&
               n_rays < (1ll << 25) && h_params->n_steps <= 128;
    if (h_params->coarse_tau < 0.f || h_params->coarse_tau > 1.f) return NEFII_E_ARG;
    P.tau = J.coarse ? h_params->coarse_tau : 0.f;
    P.cap = coarse_cap(h_params);
    P.chunk = 0;
    P.chunk_gate = 1e30f;
    P.window = 0;
    P.tier_band = 0.f;
    P.tier_gate = 0.f;
    if (h_params->trace_tier < 0 || h_params->trace_tier > 1 || h_params->tier_kappa < 0.f || h_params->tier_gate < 0.f)
        return NEFII_E_ARG;
    if (J.coarse && h_params->trace_tier) {
        // the band must cover what the recurrence decides: v <= thr and v < 0, given |v16 - v| < tau
        const float kappa = h_params->tier_kappa > 0.f ? h_params->tier_kappa : 2.f;
        const float band = kappa * P.tau, least = P.tau + 2.f * fabsf(h_params->sdf_threshold);
        P.tier_band = band > least ? band : least;
        P.tier_gate = (h_params->tier_gate > 0.f ? h_params->tier_gate : 4.f) * P.tau;
    }
    if (J.coarse && h_params->n_steps >= 16) {      // NEFII_SAMPLER_WINDOW=0: whole rows (A/B switch)
        // bit 0: quarter rows, bit 1: two-stage min-SDF refinement.  Both trade rounds for evaluations: the two-stage
        // refinement pays everywhere (config 2: 3.05 -> 2.95 ms per step), the quarter rows' three extra rounds only where
        // evaluations, not round latency, make the trace (config 3: -4 %; config 2's 4096 rays: +4 %)
        const char *e = getenv("NEFII_SAMPLER_WINDOW");
        P.window = e ? (atoi(e) & 3) : (2 | (n_rays >= 32768 ? 1 : 0));
    }
    if (J.coarse) {       // NEFII_SAMPLER_CHUNK: leading samples of a bracket search evaluated exactly first (0: off; A/B switch)
        const char *e = getenv("NEFII_SAMPLER_CHUNK");
        const int c = e ? atoi(e) : 6;       // 4 / 6 / 8 / 16: 195.8 / 196.9 / 197.2 / 203.7 ms per step on config 3 (208.7 without), 2.82 / 2.79 / 2.79 / 2.81 on config 2
        P.chunk = (c >= 2 && c <= 31 && c <= P.cap && c < h_params->n_steps) ? c : 0;
        const char *g = getenv("NEFII_SAMPLER_CHUNK_GATE");
        // 1: the chunk's last sample is within reach of the surface if |grad sdf| <= 1.  Config 3 / 4, ms per step: no gate
        // 197.3 / 174.0, gate 4: 196.7 / 173.2, 2: 195.5 / 172.0, 1: 195.1 / 171.6, 0.7: 193.9 / 171.1, 0.35: 194.2 / 171.5
        P.chunk_gate = g ? (float)atof(g) : 1.0f;
    }
    if (h_params->minsdf_lipschitz < 0.f || h_params->minsdf_lipschitz > 1e6f) return NEFII_E_ARG;
    const bool staged = J.coarse && minsdf_staged(h_params);
    P.lip = staged ? h_params->minsdf_lipschitz : 0.f;
    {       // NEFII_BRACKET_STAGED=0: the bracket search keeps its quarter-row windows (A/B switch)
        static const int v = [] {
            const char *e = getenv("NEFII_BRACKET_STAGED");
            return e ? atoi(e) != 0 : 1;
        }();
        P.stage_bracket = v;
    }
    if (h_params->unread_misses < 0 || h_params->unread_misses > 1) return NEFII_E_ARG;
    P.miss_argmin = !(h_params->unread_misses && !h_params->training);
    // (the workspace is laid out by the PARAMETERS, as nefii_trace_workspace_bytes sized it, whether or not the net takes the coarse pass)
    const int64_t step_rows = minsdf_staged(h_params) ? minsdf_rows(n_rays, &P.p) : 0;
    size_t off = carve(P.s, (char *)workspace, n_rays, h_params->n_steps, P.cap, step_rows);
    P.counters = (int *)((char *)workspace + off);
    P.levels = levels;
    P.tri_nodes = (1 << levels) - 1;
    if (reset) {      // a continuation keeps the ray state and counters in the workspace
        hipError_t e = hipMemsetAsync(P.counters, 0, sizeof(int) * NCNT * J.rounds, J.st);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(P.s.flags, 0, sizeof(int) * n_rays, J.st);
        if (e != hipSuccess) return (int)e;
        if (staged && h_params->training) {
            hipLaunchKernelGGL(minsdf_order_kernel, dim3((int)minsdf_rows(n_rays, &P.p)), dim3(128), 0, J.st, P);
            HIP_CHECK_LAUNCH();
        }
    }
    J.adv_blocks = (int)((n_rays + 255) / 256);
    // eval grid: enough workgroups for the largest possible round, capped at 2 per CU (grid-stride beyond)
    const int64_t max_q = n_rays * (int64_t)h_params->n_steps;   // n_steps >= 2^levels > bisection tree nodes > 2 ends
    const int64_t max_tiles = (max_q + TILE - 1) / TILE;
    J.eval_blocks = (int)(max_tiles < 1024 ? max_tiles : 1024);
    const int64_t max_tiles_w = (max_q + TILE_W - 1) / TILE_W;
    // tile evaluators: up to 2048 workgroups (one resident per CU: LDS), i.e. one or two tiles each for the rounds of the
    // headline workloads - the hardware dispatcher then balances tiles over CUs, also between the kernels of two traces in
    // flight (measured 256 / 512 / 768 / 2048: 4.73 / 4.68 / 4.58.. / -3 % ms per step on config 2, flat on config 3)
    static const int grid_cap = [] {
        const char *e = getenv("NEFII_EVAL_GRID");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 2048;
    }();
    J.eval_blocks_w = (int)(max_tiles_w < grid_cap ? max_tiles_w : grid_cap);
    return 0;
}

int launch_round(const TraceJob &J, int r, bool profile) {
    hipStream_t st = J.st;
    hipLaunchKernelGGL(advance_kernel, dim3(J.adv_blocks), dim3(256), 0, st, J.P, r);
    HIP_CHECK_LAUNCH();
    if (r + 1 < J.rounds) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) {
            e0 = prof_event();
            e1 = prof_event();
            (void)hipEventRecord(e0, st);
        }
        if (J.precision == 2 && J.pipelined == 2) {
            hipLaunchKernelGGL((eval_kernel16q<6, 2>), dim3(J.eval_blocks_w), dim3(512), 0, st, J.P, *J.sdf, r);
            HIP_CHECK_LAUNCH();
            const int64_t small_tiles = (J.P.n * 2 + 31) / 32 < 256 ? (J.P.n * 2 + 31) / 32 : 256;
            hipLaunchKernelGGL((eval_kernel16q<2, 2>), dim3((int)(small_tiles < 1 ? 1 : small_tiles)), dim3(512), 0, st, J.P,
                               *J.sdf, r);
        } else if (J.precision == 2 && J.pipelined && J.sdf->reserved == 1 && J.f8) {
            hipLaunchKernelGGL((eval_kernel16f<4>), dim3(J.eval_blocks_w), dim3(512), 0, st, J.P, *J.sdf, r, J.f8);
            HIP_CHECK_LAUNCH();
            const int64_t small_tiles = (J.P.n * 2 + 31) / 32 < 256 ? (J.P.n * 2 + 31) / 32 : 256;
            hipLaunchKernelGGL((eval_kernel16f<2>), dim3((int)(small_tiles < 1 ? 1 : small_tiles)), dim3(512), 0, st, J.P, *J.sdf,
                               r, J.f8);
        } else if (J.precision == 2 && J.pipelined && J.sdf->reserved == 1) {
            hipLaunchKernelGGL((eval_kernel16q<4, 4>), dim3(J.eval_blocks_w), dim3(512), 0, st, J.P, *J.sdf, r);
            HIP_CHECK_LAUNCH();
            const int64_t small_tiles = (J.P.n * 2 + 31) / 32 < 256 ? (J.P.n * 2 + 31) / 32 : 256;
            if (J.P.n <= 1024)
                hipLaunchKernelGGL((eval_kernel16q<2, 4, true>), dim3((int)(small_tiles < 1 ? 1 : small_tiles)), dim3(512), 0,
                                   st, J.P, *J.sdf, r);
            else
                hipLaunchKernelGGL((eval_kernel16q<2, 4>), dim3((int)(small_tiles < 1 ? 1 : small_tiles)), dim3(512), 0, st,
                                   J.P, *J.sdf, r);
        } else if (J.precision == 2 && J.pipelined) {
            hipLaunchKernelGGL((eval_kernel16p<P16W, 2>), dim3(J.eval_blocks_w), dim3(64 * P16W), 0, st, J.P, *J.sdf, r);
            HIP_CHECK_LAUNCH();
            const int64_t small_tiles = (J.P.n * 2 + 31) / 32 < 256 ? (J.P.n * 2 + 31) / 32 : 256;   // >= SMALL_ROUND / 32
            hipLaunchKernelGGL((eval_kernel16p<P16W, 1>), dim3((int)(small_tiles < 1 ? 1 : small_tiles)), dim3(64 * P16W),
                               0, st, J.P, *J.sdf, r);
        }
        else if (J.precision == 2)
            hipLaunchKernelGGL(eval_kernel16w, dim3(J.eval_blocks_w), dim3(WG_W), 0, st, J.P, *J.sdf, r);
        else if (J.precision == 1)
            hipLaunchKernelGGL(eval_kernel16, dim3(J.eval_blocks), dim3(WG), 0, st, J.P, *J.sdf, r);
        else
            hipLaunchKernelGGL(eval_kernel, dim3(J.eval_blocks), dim3(WG), 0, st, J.P, *J.sdf, r);
        HIP_CHECK_LAUNCH();
        if (J.coarse) {
            const int ft = J.pipelined == 2 ? 2 : 4, rows = coarse_rows(ft);
            const int64_t t = (J.P.n * (int64_t)(4 * coarse_window(J.P.p.n_steps) + 2) + rows - 1) / rows;
            const dim3 grid((int)(t < J.eval_blocks_w ? t : J.eval_blocks_w));
            NEFII_COARSE_LAUNCH(eval_kernel16s, eval_kernel16d, ft, grid, st, J.P, *J.sdf, r);
            HIP_CHECK_LAUNCH();
        }
        if (profile) (void)hipEventRecord(e1, st);
    }
    return 0;
}

int finish_job(const TraceJob &J) {
    if (J.counters) {
        hipError_t e = hipMemcpyAsync(J.counters, J.P.counters, sizeof(int) * NCNT * J.rounds, hipMemcpyDeviceToDevice, J.st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" int nefii_trace_rays_rounds(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params,
                                       const float *origins, const float *dirs, const uint8_t *object_mask,
                                       int64_t n_rays, const float *lin_steps, const float *minsdf_steps,
                                       float *out_points, uint8_t *out_hit, float *out_dists, void *workspace,
                                       size_t workspace_bytes, int32_t *counters, int round_begin, int round_end,
                                       void *stream) {
    if (n_rays == 0 && h_sdf && h_params) return 0;
    TraceJob J;
    int rc = prepare_job(J, h_sdf, h_params, origins, dirs, object_mask, n_rays, lin_steps, minsdf_steps, out_points,
                         out_hit, out_dists, workspace, workspace_bytes, counters, round_begin == 0, stream);
    if (rc) return rc;
    if (round_end <= 0 || round_end > J.rounds) round_end = J.rounds;
    if (round_begin < 0 || round_begin >= round_end) return NEFII_E_ARG;
    if (g_prof.on) {
        if (!g_prof.t0) {
            (void)hipEventCreate(&g_prof.t0);
            (void)hipEventCreate(&g_prof.t1);
        }
        (void)hipEventRecord(g_prof.t0, J.st);
    }
    for (int r = round_begin; r < round_end; ++r) {
        rc = launch_round(J, r, g_prof.on);
        if (rc) return rc;
    }
    if (g_prof.on) {
        (void)hipEventRecord(g_prof.t1, J.st);
        g_prof.have_span = true;
    }
    return finish_job(J);
}

extern "C" int nefii_trace_rays_groups(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params,
                                       const float *origins, const float *dirs, const uint8_t *object_mask,
                                       int n_groups, const int64_t *group_begin, const float *lin_steps,
                                       const float *minsdf_steps, float *out_points, uint8_t *out_hit,
                                       float *out_dists, void *const *workspaces, const size_t *workspace_bytes,
                                       int32_t *counters, int round_begin, int round_end, void *const *streams) {
    if (n_groups < 1 || n_groups > 16 || !group_begin || !workspaces || !workspace_bytes || !streams) return NEFII_E_ARG;
    if (!h_params) return NEFII_E_ARG;
    TraceJob J[16];
    const int rounds = nefii_trace_max_rounds(h_params);
    if (round_end <= 0 || round_end > rounds) round_end = rounds;
    if (round_begin < 0 || round_begin >= round_end) return NEFII_E_ARG;
    for (int g = 0; g < n_groups; ++g) {
        const int64_t lo = group_begin[g], n = group_begin[g + 1] - lo;
        if (lo < 0 || n <= 0) return NEFII_E_ARG;
        int rc = prepare_job(J[g], h_sdf, h_params, origins + 3 * lo, dirs + 3 * lo, object_mask + lo, n, lin_steps,
                             minsdf_steps, out_points + 3 * lo, out_hit + lo, out_dists + lo, workspaces[g],
                             workspace_bytes[g], counters ? counters + (size_t)g * rounds * NCNT : nullptr,
                             round_begin == 0, streams[g]);
        if (rc) return rc;
    }
    for (int r = round_begin; r < round_end; ++r)       // round-major: every chunk advances together
        for (int g = 0; g < n_groups; ++g) {
            int rc = launch_round(J[g], r, false);
            if (rc) return rc;
        }
    for (int g = 0; g < n_groups; ++g) {
        int rc = finish_job(J[g]);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int nefii_trace_rays(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params, const float *origins,
                                const float *dirs, const uint8_t *object_mask, int64_t n_rays, const float *lin_steps,
                                const float *minsdf_steps, float *out_points, uint8_t *out_hit, float *out_dists,
                                void *workspace, size_t workspace_bytes, int32_t *counters, void *stream) {
    return nefii_trace_rays_rounds(h_sdf, h_params, origins, dirs, object_mask, n_rays, lin_steps, minsdf_steps,
                                   out_points, out_hit, out_dists, workspace, workspace_bytes, counters, 0, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// camera rays (rend_util.py:90-142)
// ------------------------------------------------------------------------------------------------
__global__ void camera_rays_kernel(const float *__restrict__ uv, const float *__restrict__ pose,
                                   const float *__restrict__ K, int batch, int64_t samples, float *__restrict__ dirs,
                                   float *__restrict__ origins) {
    const int64_t total = (int64_t)batch * samples;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / samples);
        const float *Kb = K + b * 16, *Pb = pose + b * 16;
        const float fx = Kb[0], sk = Kb[1], cx = Kb[2], fy = Kb[5], cy = Kb[6];
        const float u = uv[i * 2], v = uv[i * 2 + 1];
        // x = (u - cx + cy*sk/fy - sk*v/fy) / fx * z ; y = (v - cy)/fy * z ; z = 1   (lift, :129-142)
        const float x = __fdiv_rn(fsub(fadd(fsub(u, cx), __fdiv_rn(fmul(cy, sk), fy)), __fdiv_rn(fmul(sk, v), fy)), fx);
        const float y = __fdiv_rn(fsub(v, cy), fy);
        float w[3];
        for (int c = 0; c < 3; ++c) {
            // world = pose * [x, y, 1, 1]; dirs = world - cam
            const float wc = fadd(fadd(fadd(fmul(Pb[c * 4], x), fmul(Pb[c * 4 + 1], y)), Pb[c * 4 + 2]), Pb[c * 4 + 3]);
            w[c] = fsub(wc, Pb[c * 4 + 3]);
        }
        const float nrm = fmaxf(sqrtf(fadd(fadd(fmul(w[0], w[0]), fmul(w[1], w[1])), fmul(w[2], w[2]))), 1e-12f);
        for (int c = 0; c < 3; ++c) {
            dirs[i * 3 + c] = __fdiv_rn(w[c], nrm);
            origins[i * 3 + c] = Pb[c * 4 + 3];
        }
    }
}

extern "C" int nefii_camera_rays(const float *uv, const float *pose, const float *intrinsics, int batch,
                                 int64_t samples, float *out_dirs, float *out_origins, void *stream) {
    if (!uv || !pose || !intrinsics || !out_dirs || !out_origins) return NEFII_E_ARG;
    const int64_t total = (int64_t)batch * samples;
    if (total <= 0) return 0;
    int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(camera_rays_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, uv, pose, intrinsics, batch,
                       samples, out_dirs, out_origins);
    HIP_CHECK_LAUNCH();
    return 0;
}
