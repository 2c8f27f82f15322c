// Device-side data structures and kernels of libsvicp (gfx950 only).
//
// Layout in HBM
//   target   : float4 {x, y, z, original index as int bits}, sorted by linear
//              cell key (z-major, x fastest), so the points of x-adjacent cells
//              are contiguous;  normals packed 3 x fp32 in the same order.
//   grid     : cells of edge h >= 2 * search radius.  Cells are grouped in
//              4x4x4 blocks; `dir` maps a block to its rank among the blocks
//              that hold (or precede in x) a point, -1 otherwise.  Per ranked
//              block 64 entries: run2 = {first point of the cell, end of the
//              NEXT cell in x} and end1 = end of the cell itself; quad = per
//              quarter cell the four runs a probe that starts there walks (its
//              own row of cells and the rows next to it in y, in z and in
//              both), packed in ONE 16-byte word (one table look-up and one
//              load per probe instead of four and four); the rare record that
//              does not fit the packing is kept in full in quad_big.
//   source   : SoA x[], y[], z[] in Morton order (perm[] maps back).
//
// A query probes the 2x2x2 cells that cover its search cube: four 2-cell runs.
// A block is ranked when a probe that starts in it can meet a point: it holds a
// point, or a cell one step below a point's cell in x, y and/or z.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svicp {

// Development builds (-DSVICP_TRACE) stamp the phases of every block of the iteration
// kernel with the 100 MHz wall clock; svicp_debug_trace() copies the stamps out.
#ifdef SVICP_TRACE
constexpr int kTraceIters = 64, kTraceTiles = 1024, kTraceStamps = 16;
__device__ unsigned long long g_trace[kTraceIters * kTraceTiles * kTraceStamps];
#define SVICP_STAMP(k)                                                                    \
    do {                                                                                  \
        if (threadIdx.x == 0 && iter < kTraceIters && tile < kTraceTiles)                 \
            g_trace[((size_t)iter * kTraceTiles + tile) * kTraceStamps + (k)] = wall_clock64(); \
    } while (0)
#define SVICP_SOLVE_STAMP(k)                                                              \
    do {                                                                                  \
        if (t == 0 && trace_row) trace_row[k] = wall_clock64();                            \
    } while (0)
#else
#define SVICP_STAMP(k) do { } while (0)
#define SVICP_SOLVE_STAMP(k) do { } while (0)
#endif

// Development builds for the byte budget (-DSVICP_DROP=<bits>): each bit takes ONE stream of
// memory traffic out of the search and the iteration kernel, so that counter passes over such
// builds tell how many bytes each array moves (tools/byte_budget.sh).  Results of such a
// build mean nothing; the product is built without the macro and none of this is compiled in.
#ifndef SVICP_DROP
#define SVICP_DROP 0
#endif
constexpr unsigned kDrop = SVICP_DROP;
constexpr int kDropIter = 2;    // the iteration kernel drops its streams in this launch of a run only
constexpr unsigned kDropSrcWrite = 1,    // the updated coordinates are not stored
                   kDropStateWrite = 2,  // the per-point state (28 B) is not stored
                   kDropNrm = 4,         // the winner's normal is not gathered
                   kDropPts = 8,         // candidates are made up instead of gathered
                   kDropQuad = 16,       // the probe record is not loaded (no candidates either)
                   kDropDir = 32,        // nor the directory word
                   kDropOut = 64;        // nn1_kernel: index and distance are not stored

// A/B builds (-DSVICP_LATE_STATE=k): the per-point state is first written in launch k of a
// run and first read in launch k + 1 (what a policy that skips the state while the cloud still
// moves far would do on the benchmark pair with k = 2).
#ifndef SVICP_LATE_STATE
#define SVICP_LATE_STATE 0
#endif
constexpr int kLateState = SVICP_LATE_STATE;

constexpr int kBlockThreads = 256;
constexpr int kNumSums = 32;   // 21 JtJ + 6 Jtr + inliers + sum w b^2 + sum w + 2 pad
// Same-address fp64 atomics serialise in L2 (about 6 ns each): the per-iteration change of
// the sums is therefore kept in kDeltaCopies copies, a tile adds to copy (tile mod copies),
// and the solve adds the copies up in a fixed order.
constexpr int kDeltaCopies = 4;   // measured: 1 copy costs 9 us per search-heavy iteration, 16 slow the solve
constexpr int kDeltaRow = kDeltaCopies * kNumSums;   // doubles per iteration
// Relative slack of the proof that a remembered correspondence still holds.  A computed
// fp32 squared distance is within 5 * 2^-24 = 3e-7 of the true one and a computed square root
// within 2^-24: every bound is relaxed by kSlack on the distance (three times that on a
// squared distance), several times what rounding can add up to.
#ifndef SVICP_SLACK
#define SVICP_SLACK 1e-6f
#endif
constexpr float kSlackUp = 1.0f + SVICP_SLACK, kSlackDown = 1.0f - SVICP_SLACK,
                kSlackUp2 = 1.0f + 3.0f * SVICP_SLACK;
constexpr int kMaxRing = 8;    // exact search: rings tried before the brute-force pass
constexpr int kKnnMaxRing = 8; // the same for the k-NN facade

struct GridDesc {
    float ox, oy, oz;        // origin (one empty cell below the target's bounding box)
    float h, inv_h;          // cell edge
    float rr;                // probe radius: sqrt(threshold) plus rounding slack
    float cov_slack;         // absolute slack when a scanned box is declared to cover a radius
    float fnx1, fny1, fnz1;  // (float)(n - 1): upper bound of a valid probe origin
    int nx, ny, nz;          // cells per axis
    int nbx, nby, nbz;       // blocks per axis
    int m;                   // target points
    const int *dir;          // [nbz][nby][nbx]
    const uint2 *run2;       // [ranked blocks * 64]
    const unsigned *end1;    // [ranked blocks * 64]
    const uint4 *quad;       // [ranked blocks * 256]: the four runs a probe walks, one packed record per
                             // QUARTER cell in x (kFineX per cell): see probe_origin, probe_runs
    const uint4 *quad_big;   // [slots][2]: the records that do not fit the packing, in full
    const float4 *pts;       // [m] sorted
    const float *nrm;        // [3 m] sorted
};

// Everything the iteration loop keeps on the device between kernels.
struct IcpState {
    double R[9];
    double T[3];
    double Trans[16];
    double S[kNumSums];      // last reduced sums
    int status;
    int pad;
};

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool better(float d, int id, float bd, int bid)
{
    return (d < bd) || (d == bd && id < bid);
}

// Squared distance exactly as flann::L2<float> accumulates it for three
// dimensions: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to fp32.
__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4 &p)
{
    float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    return d;
}

__device__ __forceinline__ int cell_local(int cx, int cy, int cz)
{
    return ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
}

__device__ __forceinline__ int dir_index(const GridDesc &g, int cx, int cy, int cz)
{
    return ((cz >> 2) * g.nby + (cy >> 2)) * g.nbx + (cx >> 2);
}

// Probe origin: the cell that holds q - rr on each axis.  Returns false when
// the 2x2x2 probe would leave the grid, i.e. no target point lies within rr.
//
// The target is sorted by cell key with x fastest, and inside a cell by the QUARTER of the
// cell its x falls in (kFineX quarters): cells next to each other in x, and quarters inside
// them, are contiguous.  The cube [q - rr, q + rr] is less than a cell wide, i.e. less than
// four quarters: it lies in the quarter of q - rr and the four after it, 1.25 cells instead
// of the 2 whole cells a probe by cells has to take -- 0.63 times the candidates for the same
// four runs (8.0 instead of 12.7 per query on the benchmark pair, 3.0 instead of 4.2 chunks
// of four per wave).  cxq: the quarter's number along x (4 cx + 0..3); scaling by four is
// exact, so cxq >> 2 is the cell the unscaled quotient gives.
#ifndef SVICP_FINEX
#define SVICP_FINEX 4
#endif
constexpr int kFineX = SVICP_FINEX;     // a power of two (4: quarter cells; other values for A/B builds)
constexpr int kFineRun = kFineX + 1;    // quarters a probe's run spans
constexpr int kFineBits = kFineX == 1 ? 2 : kFineX == 2 ? 3 : kFineX == 4 ? 4 : 5;   // log2 of the quarters in a block's row
static_assert((1 << kFineBits) == 4 * kFineX, "kFineX is 1, 2, 4 or 8");
__device__ __forceinline__ bool probe_origin(const GridDesc &g, float qx, float qy, float qz,
                                             int &cx, int &cy, int &cz, int *cxq = nullptr)
{
    float fx = ((qx - g.rr) - g.ox) * g.inv_h;
    float fy = ((qy - g.rr) - g.oy) * g.inv_h;
    float fz = ((qz - g.rr) - g.oz) * g.inv_h;
    bool ok = (fx >= 0.f) && (fy >= 0.f) && (fz >= 0.f) && (fx < g.fnx1) && (fy < g.fny1) &&
              (fz < g.fnz1);
    if (!ok) return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    if (cxq) *cxq = (int)(fx * (float)kFineX);
    return true;
}

// Place of quarter cell (cxq, cy, cz) among the 256 of its block of 4 x 4 x 4 cells.
__device__ __forceinline__ int quarter_local(int cxq, int cy, int cz)
{
    return ((cz & 3) << (kFineBits + 2)) | ((cy & 3) << kFineBits) | (cxq & (4 * kFineX - 1));
}

// A probe record: the four runs {start, end} a probe from one quarter cell walks, run r in
// the row of cells (cy + (r & 1), cz + (r >> 1)).  Packed in 16 bytes: x = start of run 0,
// y = start of run 2, z = (start of run 1 - start of run 0) | (start of run 3 - start of
// run 2) << 16, w = the four lengths, a byte each.  (Rows next to each other in y follow
// each other in the sorted target, so the differences are small; an empty run is given its
// partner's start.  Half the bytes of eight plain words: a launch that searches every point
// moves a third of its bytes for these records, and it runs at the speed of its bytes.)  A
// record with a length above 254 or a difference above 65535 -- duplicates piled up in a cell,
// rows of more than 65535 points -- has w = kQuadBig and in x its slot in quad_big, where it
// is kept as eight plain words.
constexpr unsigned kQuadBig = 0xffffffffu;
__device__ __forceinline__ void probe_runs(const GridDesc &g, size_t e, uint2 (&run)[4])
{
    const uint4 c = g.quad[e];
    // (all four words are used before the branch: the record stays one 16-byte load, and the
    // packed form's runs are ready without waiting for the branch)
    const unsigned s1 = c.x + (c.z & 0xffffu), s3 = c.y + (c.z >> 16);
    run[0] = make_uint2(c.x, c.x + (c.w & 0xffu));
    run[1] = make_uint2(s1, s1 + ((c.w >> 8) & 0xffu));
    run[2] = make_uint2(c.y, c.y + ((c.w >> 16) & 0xffu));
    run[3] = make_uint2(s3, s3 + (c.w >> 24));
    if (__builtin_expect(c.w == kQuadBig, 0)) {
        const uint4 a = g.quad_big[2 * (size_t)c.x], b = g.quad_big[2 * (size_t)c.x + 1];
        run[0] = make_uint2(a.x, a.y); run[1] = make_uint2(a.z, a.w);
        run[2] = make_uint2(b.x, b.y); run[3] = make_uint2(b.z, b.w);
    }
}

// Walks four runs of candidates as one flat list, CH at a time.  `fetch(j)` returns point j
// {x, y, z, original index} from wherever the caller keeps it (global memory, LDS).
template <int CH, bool SECOND, class Fetch>
__device__ __forceinline__ void scan_runs(const uint2 (&run)[4], float qx, float qy, float qz,
                                          Fetch &&fetch, float &best, int &bpos, int &bidx,
                                          float *second)
{
    constexpr int C = CH;
    static_assert(C >= 2 && C % 2 == 0, "candidates are evaluated in pairs");
    const unsigned P1 = run[0].y - run[0].x;
    const unsigned P2 = P1 + (run[1].y - run[1].x);
    const unsigned P3 = P2 + (run[2].y - run[2].x);
    const unsigned T = P3 + (run[3].y - run[3].x);
    const unsigned o0 = run[0].x, o1 = run[1].x - P1, o2 = run[2].x - P2, o3 = run[3].x - P3;
    // (distance bits, index) as one 64-bit key: distances are >= +0, so their
    // bit patterns order like the values and one unsigned compare applies the
    // rule "smaller distance, then smaller index".  A slot past the end gets
    // distance +inf, which never beats the initial key.
    unsigned long long bkey = (unsigned long long)0x7f800000u << 32;
    float b1 = INFINITY, b2 = INFINITY;   // two smallest distances (SECOND only)
    const v2f q2x = { qx, qx }, q2y = { qy, qy }, q2z = { qz, qz };
    for (unsigned f = 0; f < T; f += C) {
        float4 p[C];
        unsigned jj[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            unsigned ff = f + c;
            unsigned j = ff + (ff < P1 ? o0 : ff < P2 ? o1 : ff < P3 ? o2 : o3);
            jj[c] = ff < T ? j : 0u;     // past the end: re-read point 0, result dropped
            p[c] = fetch(jj[c]);
        }
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            // two candidates per packed fp32 instruction; each element is the
            // same separately rounded ((dx*dx)+dy*dy)+dz*dz as dist2()
            v2f dx = q2x - (v2f){ p[c].x, p[c + 1].x };
            v2f dy = q2y - (v2f){ p[c].y, p[c + 1].y };
            v2f dz = q2z - (v2f){ p[c].z, p[c + 1].z };
            v2f d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float de = (f + c + e < T) ? (e ? d.y : d.x) : INFINITY;
                unsigned long long key = ((unsigned long long)__float_as_uint(de) << 32) |
                                         __float_as_uint(p[c + e].w);
                bool take = key < bkey;
                bkey = take ? key : bkey;
                bpos = take ? (int)jj[c + e] : bpos;
                if (SECOND) {
                    b2 = fminf(b2, fmaxf(de, b1));
                    b1 = fminf(b1, de);
                }
            }
        }
    }
    if (bpos >= 0) {
        best = __uint_as_float((unsigned)(bkey >> 32));
        bidx = (int)(unsigned)bkey;
    }
    if (SECOND) *second = b2;
}

// Nearest target point within the probe radius: the 1-NN the ICP loop needs.
// best = +inf, bpos = -1 when the probed cells are empty.
//
// The four runs are walked as one flat candidate list, CH candidates at a
// time: the CH loads of a chunk are issued together (one memory latency per
// chunk instead of one per candidate) and a wave iterates max-over-lanes of
// the TOTAL candidate count, not the sum of per-run maxima.
template <int CH, bool SECOND = false>
__device__ __forceinline__ void nn_capped(const GridDesc &g, float qx, float qy, float qz,
                                          float &best, int &bpos, int &bidx, float *second = nullptr,
                                          const bool dropping = true)
{
    best = INFINITY; bpos = -1; bidx = 0x7fffffff;
    if (SECOND) *second = INFINITY;
    int cx, cy, cz, cxq;
    if (!probe_origin(g, qx, qy, qz, cx, cy, cz, &cxq)) return;

    // run r covers the quarter cells cxq .. cxq + 4 of the row (cy + (r & 1), cz + (r >> 1))
    const int rank = ((kDrop & kDropDir) && dropping) ? (cx + cy + cz) % max(g.m / 64, 1) : g.dir[dir_index(g, cx, cy, cz)];
    // an unranked block has no entry: read entry 0 of the table and drop it
    const size_t e = rank >= 0 ? (size_t)rank * (64 * kFineX) + quarter_local(cxq, cy, cz) : 0;
    uint2 run[4];
    if ((kDrop & (kDropQuad | kDropDir)) && dropping) {
        // made-up runs of the usual lengths (2 + 2 + 2 + 2) at a place that depends on the query
        const unsigned s0 = (unsigned)((cx * 7 + cy * 131 + cz * 977) & 0xffff);
#pragma unroll
        for (int r = 0; r < 4; ++r) run[r] = make_uint2(s0 + 16u * r, s0 + 16u * r + 2u);
    } else {
        probe_runs(g, e, run);
    }
    if (rank < 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) run[r] = make_uint2(0u, 0u);
    }
    if ((kDrop & (kDropPts | kDropQuad | kDropDir)) && dropping)
        scan_runs<CH, SECOND>(run, qx, qy, qz,
                              [&](unsigned j) { return make_float4(qx + 0.001f * (float)(j & 1023u), qy, qz, __uint_as_float(j)); },
                              best, bpos, bidx, second);
    else
        scan_runs<CH, SECOND>(run, qx, qy, qz, [&](unsigned j) { return g.pts[j]; }, best, bpos, bidx,
                              second);
}

// The same search by one whole wave for ONE query (all lanes pass the same q): the
// candidates are spread over the lanes, and a butterfly picks the winner.  Same candidates, same distance arithmetic and the same
// order rule as nn_capped, so the same answer; the chain of dependent loads is table,
// run, candidates instead of one load per chunk of candidates.  `bpt` is the winning
// target point.  Results are uniform over the wave.
template <bool SECOND>
__device__ __forceinline__ void nn_capped_wave(const GridDesc &g, float qx, float qy, float qz,
                                               float &best, int &bpos, int &bidx, float4 &bpt,
                                               float *second)
{
    best = INFINITY; bpos = -1; bidx = 0x7fffffff;
    bpt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SECOND) *second = INFINITY;
    int cx, cy, cz, cxq;
    if (!probe_origin(g, qx, qy, qz, cx, cy, cz, &cxq)) return;
    const int lane = threadIdx.x & 63;
    const int rank = g.dir[dir_index(g, cx, cy, cz)];       // the same address in every lane
    if (rank < 0) return;
    const size_t e = (size_t)rank * (64 * kFineX) + quarter_local(cxq, cy, cz);
    uint2 run[4];
    probe_runs(g, e, run);
    const unsigned P1 = run[0].y - run[0].x;
    const unsigned P2 = P1 + (run[1].y - run[1].x);
    const unsigned P3 = P2 + (run[2].y - run[2].x);
    const unsigned T = P3 + (run[3].y - run[3].x);
    const unsigned o0 = run[0].x, o1 = run[1].x - P1, o2 = run[2].x - P2, o3 = run[3].x - P3;
    unsigned long long bkey = (unsigned long long)0x7f800000u << 32;
    float b1 = INFINITY, b2 = INFINITY;
    float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
    int mpos = -1;
    for (unsigned f = lane; f < T; f += 64) {
        const unsigned j = f + (f < P1 ? o0 : f < P2 ? o1 : f < P3 ? o2 : o3);
        const float4 p = g.pts[j];
        // every candidate's normal is touched in the same trip: the caller's gather of the
        // winner's normal then finds its line in the cache instead of making a trip of its own
        const float touched_normal = g.nrm[3 * (size_t)j];
        const float d = dist2(qx, qy, qz, p);
        asm volatile("" ::"v"(touched_normal));
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(p.w);
        if (key < bkey) { bkey = key; mp = p; mpos = (int)j; }
        b2 = fminf(b2, fmaxf(d, b1));
        b1 = fminf(b1, d);
    }
    unsigned long long wkey = bkey;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(wkey, off, 64);
        wkey = o < wkey ? o : wkey;
        if (SECOND) {
            const float c1 = __shfl_xor(b1, off, 64), c2 = __shfl_xor(b2, off, 64);
            b2 = fminf(fmaxf(b1, c1), fminf(b2, c2));
            b1 = fminf(b1, c1);
        }
    }
    const unsigned long long owners = __ballot(mpos >= 0 && bkey == wkey);
    if (owners == 0ull) return;                       // no candidate at all
    const int w = __ffsll((long long)owners) - 1;     // keys are distinct: one owner
    bpos = __shfl(mpos, w, 64);
    bpt.x = __shfl(mp.x, w, 64); bpt.y = __shfl(mp.y, w, 64); bpt.z = __shfl(mp.z, w, 64);
    bpt.w = __shfl(mp.w, w, 64);
    best = __uint_as_float((unsigned)(wkey >> 32));
    bidx = (int)(unsigned)wkey;
    if (SECOND) *second = b2;
}

// Asks for the lines a search from (qx, qy, qz) -- the same in every lane -- would read, and
// waits for none of the candidates: the directory word, the probe record, then lane f the
// f-th candidate and its normal (t0, t1: to be kept alive by the caller until it can afford
// to wait).  Any position is safe, a stale or never-written one included: the origin test
// rejects what lies outside the grid (NaN too), and inside it every table entry is valid.
__device__ __forceinline__ void probe_touch(const GridDesc &g, float qx, float qy, float qz, float &t0,
                                            float &t1)
{
    int cx, cy, cz, cxq;
    if (!probe_origin(g, qx, qy, qz, cx, cy, cz, &cxq)) return;
    const int rank = g.dir[dir_index(g, cx, cy, cz)];
    if (rank < 0) return;
    uint2 run[4];
    probe_runs(g, (size_t)rank * (64 * kFineX) + quarter_local(cxq, cy, cz), run);
    const unsigned P1 = run[0].y - run[0].x;
    const unsigned P2 = P1 + (run[1].y - run[1].x);
    const unsigned P3 = P2 + (run[2].y - run[2].x);
    const unsigned T = P3 + (run[3].y - run[3].x);
    const unsigned f = threadIdx.x & 63;
    if (f < T) {
        const unsigned j = f + (f < P1 ? run[0].x : f < P2 ? run[1].x - P1 : f < P3 ? run[2].x - P2 : run[3].x - P3);
        t0 = g.pts[j].x;
        t1 = g.nrm[3 * (size_t)j];
    }
}

// Visits every target point in the cell box [x0,x1] x [y0,y1] x [z0,z1]
// (already clipped to the grid) exactly once.
template <class F>
__device__ __forceinline__ void scan_box(const GridDesc &g, int x0, int x1, int y0, int y1,
                                         int z0, int z1, F &&visit)
{
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
            for (int bx = x0 >> 2; bx <= (x1 >> 2); ++bx) {
                int rank = g.dir[((z >> 2) * g.nby + (y >> 2)) * g.nbx + bx];
                if (rank < 0) continue;
                int xa = max(x0, bx * 4), xb = min(x1, bx * 4 + 3);
                int base = rank * 64 + (((z & 3) << 4) | ((y & 3) << 2));
                unsigned s = g.run2[base + (xa & 3)].x;
                unsigned e = g.end1[base + (xb & 3)];
                for (unsigned j = s; j < e; ++j) visit(j);
            }
}

// The same walk with the points loaded four at a time and handed to `visit(j, point)`: the
// loads of a group are in flight together, one trip to memory per four candidates instead of
// one per candidate (a place past the end of the run reads the run's last point again, which
// changes nothing under a strict order).
template <class F>
__device__ __forceinline__ void scan_box_points(const GridDesc &g, int x0, int x1, int y0, int y1,
                                                int z0, int z1, F &&visit)
{
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
            for (int bx = x0 >> 2; bx <= (x1 >> 2); ++bx) {
                int rank = g.dir[((z >> 2) * g.nby + (y >> 2)) * g.nbx + bx];
                if (rank < 0) continue;
                int xa = max(x0, bx * 4), xb = min(x1, bx * 4 + 3);
                int base = rank * 64 + (((z & 3) << 4) | ((y & 3) << 2));
                unsigned s = g.run2[base + (xa & 3)].x;
                unsigned e = g.end1[base + (xb & 3)];
                for (unsigned j = s; j < e; j += 4) {
                    const unsigned j1 = min(j + 1, e - 1), j2 = min(j + 2, e - 1), j3 = min(j + 3, e - 1);
                    const float4 p0 = g.pts[j], p1 = g.pts[j1], p2 = g.pts[j2], p3 = g.pts[j3];
                    visit(j, p0); visit(j1, p1); visit(j2, p2); visit(j3, p3);
                }
            }
}

// Box of ring `ring` around the probe origin, clipped; returns the radius
// around q that the box is certain to cover (+inf once it spans the grid).
__device__ __forceinline__ float ring_box(const GridDesc &g, float qx, float qy, float qz, int cx,
                                          int cy, int cz, int ring, int &x0, int &x1, int &y0,
                                          int &y1, int &z0, int &z1)
{
    float rho = INFINITY;
    auto axis = [&](float q, float o, int c, int n, int &a0, int &a1) {
        int lo = c - ring, hi = c + 1 + ring;
        if (lo > 0) rho = fminf(rho, q - (o + (float)lo * g.h));
        if (hi < n - 1) rho = fminf(rho, (o + (float)(hi + 1) * g.h) - q);
        a0 = max(lo, 0);
        a1 = min(hi, n - 1);
    };
    axis(qx, g.ox, cx, g.nx, x0, x1);
    axis(qy, g.oy, cy, g.ny, y0, y1);
    axis(qz, g.oz, cz, g.nz, z0, z1);
    if (rho == INFINITY) return INFINITY;
    return rho * (1.0f - 1e-5f) - g.cov_slack;
}

__device__ __forceinline__ void clamped_origin(const GridDesc &g, float qx, float qy, float qz,
                                               int &cx, int &cy, int &cz)
{
    auto cl = [&](float q, float o, int n) {
        float f = ((q - g.rr) - o) * g.inv_h;
        f = fminf(fmaxf(f, 0.f), (float)(n - 2));   // NaN -> 0
        return (int)f;
    };
    cx = cl(qx, g.ox, g.nx); cy = cl(qy, g.oy, g.ny); cz = cl(qz, g.oz, g.nz);
}

// ------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the 64 lanes by data-parallel-primitive moves (no LDS traffic):
// four shifts inside each row of 16 lanes, then the row totals are passed on
// to the next rows.  The total is valid in lane 63 only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row total
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------
// The per-iteration kernel: transform, correspondence search, residual,
// Jacobian row and normal-equation accumulation in one pass over the source.
//
// Restates CPoint2PlaneICP.cpp:42-50 (update, fp64 math rounded to fp32),
// :65-79 (1-NN, weight) and :85-109 (row assembly, weighted normal equations).
// ------------------------------------------------------------------------
constexpr int kItThreads = 256;                  // threads per block of the iteration kernels
constexpr int kItPoints = 2;                     // (kItPoints * kItHalves points per thread)
constexpr int kItHalves = 2;                     // passes of kItPoints per tile
constexpr int kItTile = kItThreads * kItPoints * kItHalves;   // source points per block

// Row assembly and normal-equation terms of one correspondence
// (CPoint2PlaneICP.cpp:68-79 weight, :85-99 row, :101-109 weighted products).
__device__ __forceinline__ void accumulate_point(double (&acc)[30], float px, float py, float pz,
                                                 float best, float thr, float tx, float ty,
                                                 float tz, float nx, float ny, float nz)
{
    float sl = best / thr;
    float ss = __fmul_rn(sl, sl);
    double om = 1.0 - (double)ss;
    float w = (float)__dmul_rn(om, om);
    float a0 = __fsub_rn(__fmul_rn(py, nz), __fmul_rn(pz, ny));
    float a1 = __fsub_rn(__fmul_rn(pz, nx), __fmul_rn(px, nz));
    float a2 = __fsub_rn(__fmul_rn(px, ny), __fmul_rn(py, nx));
    float b = __fadd_rn(__fadd_rn(__fmul_rn(__fsub_rn(tx, px), nx),
                                  __fmul_rn(__fsub_rn(ty, py), ny)),
                        __fmul_rn(__fsub_rn(tz, pz), nz));
    double a[6] = { (double)a0, (double)a1, (double)a2, (double)nx, (double)ny, (double)nz };
    double wd = (double)w, bd = (double)b;
    int v = 0;
#pragma unroll
    // the sums are fp64 and their order is free (the reference adds them in point order);
    // fused multiply-adds halve the fp64 work of this step
    for (int r = 0; r < 6; ++r) {
        double aw = a[r] * wd;
#pragma unroll
        for (int c = r; c < 6; ++c) { acc[v] = fma(aw, a[c], acc[v]); ++v; }
        acc[21 + r] = fma(aw, bd, acc[21 + r]);
    }
    if (w > 0.f) acc[27] += 1.0;
    acc[28] = fma(wd * bd, bd, acc[28]);
    acc[29] += wd;
}

// Adds two doubles held by different lanes: the swap instruction exchanges the
// upper half of `a` with the lower half of `b` (rows of 32 lanes, or odd/even
// rows of 16 lanes), so afterwards a + b holds, in the lanes that kept `a`,
// the sum of two lanes' a, and in the other lanes the sum of two lanes' b.
template <bool ROW16>
__device__ __forceinline__ double swap_add(double a, double b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if (ROW16) {
        auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
        auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
}

// Block sum of the 30 per-thread accumulators; thread v < 32 returns total v.
// Per wave: two halving steps by lane swaps (32 values -> 16 -> 8 registers,
// each row of 16 lanes then holds 16 partial sums of one value), four shifts
// inside the rows, and the four row ends write the totals.  The order of the
// additions is fixed.
__device__ __forceinline__ double block_sum_30(double (&acc)[30], double (*red)[kNumSums])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double u[16];
#pragma unroll
    for (int j = 0; j < 15; ++j) u[j] = swap_add<false>(acc[2 * j], acc[2 * j + 1]);
    u[15] = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        double w = swap_add<true>(u[2 * m], u[2 * m + 1]);
        w = dpp_add<0x111, 0xf>(w);   // row_shr:1
        w = dpp_add<0x112, 0xf>(w);   // row_shr:2
        w = dpp_add<0x114, 0xf>(w);   // row_shr:4
        w = dpp_add<0x118, 0xf>(w);   // row_shr:8 -> lane 15 of each row holds its value's total
        // row r of register m holds value 4m + {0, 2, 1, 3}[r]
        const int row = lane >> 4;
        const int value = 4 * m + ((row & 1) << 1) + (row >> 1);
        if ((lane & 15) == 15) red[wv][value] = w;
    }
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < 30) {
#pragma unroll
        for (int w = 0; w < kItThreads / 64; ++w) s += red[w][threadIdx.x];
    }
    return s;
}

// The lower bound `lb` on a point's distance to every other target point after the point
// has moved by sqrt(m2): a float not above lb - |move| (triangle inequality).  The bound is
// carried from iteration to iteration, so what is given away here adds up: a factor
// (1 - 1e-6) per iteration wears a bound down by eight units in the last place each time,
// and a point whose two nearest target points differ by a few parts in 10^5 would fail its
// proof, and be searched again, every few iterations.  Instead: d is not below the true
// move (1e-6 covers the roundings of the three squares, their sum and the root); r = fl(lb - d)
// lies within half a unit of lb - d, and lb - r is exact when the two are within a factor of
// two of each other, so comparing it with d tells on which side; only if r came out above
// does it go down, by one unit.  (A move of more than a quarter of the bound takes the
// factor as before: nothing is carried far from there.)
__device__ __forceinline__ float bound_after_move(float lb, float m2)
{
    const float d = sqrtf(m2) * kSlackUp;
    const float r = lb - d;
    if (d > 0.25f * lb) return r * kSlackDown;
    // r > 0 here; the float below it
    return (lb - r >= d) ? r : __uint_as_float(__float_as_uint(r) - 1u);
}

// Search bounds shared by the two iteration kernels.
struct CohParams {
    float thr;        // squared-distance threshold of the weight function
    float root_hi;    // sqrt(thr) rounded up: "farther than this" means weight 0
    float cov;        // every target point within cov of a query lies in its probed cells
};

// What the loop remembers about each source point between iterations
// (arrays in source order, 28 bytes per point): the matched target point (none within cov:
// NaN), the matched normal, and lb, a lower bound on the distance from the source point to
// every OTHER target point.
struct Float3 {
    float x, y, z;    // 12 bytes, read and written as one three-dword access
};
struct PointState {
    Float3 *tp;       // matched target point; x is NaN when there is none within cov
    Float3 *nv;       // its normal
    float *lb;
    // [3][64] words of 64 bits or null: bit (t mod 64) of word (t / 64) of row (k mod 3) says
    // whether tile t had work to do in iteration k (see choose_tile).  A row is written in
    // iteration k (atomic OR by the tiles with work), read in k + 1 and cleared in k + 2.
    unsigned long long *active;
    int flag_words = 64;          // words per row of `active` (64: the 4096 tiles choose_tile handles)
    // [tiles][4 waves][kBoxWords] or null: where each wave's points lie (see box_cannot_move)
    float4 *box = nullptr;
};
constexpr int kFlagCap = 4096;   // tiles the adaptive block order handles (64 per lane of a wave)
constexpr int kFlagRows = 3, kFlagWords = kFlagCap / 64;

// Maximum / minimum over the 64 lanes, the same in every lane afterwards (moves inside rows of 16
// lanes, then the row ends passed on: no LDS traffic).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float v, float outside)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(outside), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_or<0x111, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x112, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x114, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x118, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x142, 0xa>(v, -INFINITY));
    v = fmaxf(v, dpp_or<0x143, 0xc>(v, -INFINITY));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f(float v) { return -wave_max_f(-v); }

// Per wave of a tile (64 x 4 points), two float4: {lowest x, y, z, -}, {highest x, y, z, -}; lowest
// above highest (+inf, -inf) where the wave has nothing to say.
// rounding_room(p): the step below which fl32(p + e) = p -- half a unit in the last place of p, a
// quarter where p is a power of two (the floats below it lie half as far apart), nothing where p
// is zero, tiny or not finite.  Over a set of coordinates of one sign the smallest room is that
// of the smallest magnitude (a larger one in the same binade has a mantissa, hence the half
// unit; in a higher binade even the quarter is as much).
constexpr int kBoxWords = 2;
constexpr int kBoxRecord = kBoxWords * 4;      // float4 per tile (256 threads: four waves)
__device__ __forceinline__ float rounding_room(float p)
{
    const unsigned b = __float_as_uint(p);
    const unsigned ex = b & 0x7f800000u;
    if (ex < (27u << 23) || ex == 0x7f800000u) return 0.f;
    return __uint_as_float(ex) * ((b & 0x007fffffu) ? 0x1p-24f : 0x1p-25f);
}

// Can no point inside the box [lo, hi] change under p <- fl32(R p + T)?  e = (R - I) p + T is
// bounded over the box by interval arithmetic (signs kept: near convergence the rotation's and
// the translation's shares cancel over most of the cloud, which a sum of magnitudes cannot see),
// plus a slack of 2^-45 of the magnitudes involved for the roundings of the fp64 update and of
// this test (each of a dozen operations errs by 2^-53 of its result); the point stays where it
// is if that stays strictly inside the axis's room: that of the bound nearer to zero, and none
// for a box that reaches across zero (the coordinates next to zero have next to no room anyway).
// Anything not finite fails the comparison.
__device__ __forceinline__ bool box_cannot_move(const float lo[3], const float hi[3], const double R[9], const double T[3])
{
    bool still = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double room = lo[a] > 0.f ? rounding_room(lo[a]) : hi[a] < 0.f ? rounding_room(hi[a]) : 0.f;
        double mn = T[a], mx = T[a], mag = fabs(T[a]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double e = R[3 * a + j] - (a == j ? 1.0 : 0.0);
            const double u = e * (double)lo[j], v = e * (double)hi[j];
            mn += fmin(u, v); mx += fmax(u, v);
            mag += fmax(fabs((double)lo[j]), fabs((double)hi[j]));      // |R| <= 1: the update's terms are no larger
        }
        still = still && (fmax(fabs(mn), fabs(mx)) + mag * 0x1p-45 < room);
    }
    return still;
}

// ------------------------------------------------------------------------
// 6x6 solve and transform update on the device.
// Restates CPoint2PlaneICP.cpp:109 (X), :111-119 (R = Rz*Ry*Rx, T) and
// :36-39 / :52-55 (Trans = [R T; 0 1] * Trans).
// ------------------------------------------------------------------------
// 2^-floor(e/2) for a = f * 2^e, 1 <= f < 2: a power of two near 1/sqrt(a).  Scaling rows
// and columns by it is exact and leaves the diagonal in [1, 4).
__device__ __forceinline__ double pow2_rsqrt(double a)
{
    const int e = (int)((__double_as_longlong(a) >> 52) & 0x7ff) - 1023;
    return __longlong_as_double((long long)(1023 - (e >> 1)) << 52);
}

// sin and cos of a small angle by their Taylor polynomials (|x| <= 0.5: the first
// neglected terms are 0.5^17/17! and 0.5^18/18!, far below one rounding).
__device__ __forceinline__ void sincos_small(double x, double &sn, double &cs)
{
    const double z = x * x;
    double s = -1.0 / 1307674368000.0;          // -1/15!
    s = fma(s, z, 1.0 / 6227020800.0);          //  1/13!
    s = fma(s, z, -1.0 / 39916800.0);           // -1/11!
    s = fma(s, z, 1.0 / 362880.0);              //  1/9!
    s = fma(s, z, -1.0 / 5040.0);               // -1/7!
    s = fma(s, z, 1.0 / 120.0);                 //  1/5!
    s = fma(s, z, -1.0 / 6.0);                  // -1/3!
    sn = fma(x * z, s, x);
    double c = 1.0 / 20922789888000.0;          //  1/16!
    c = fma(c, z, -1.0 / 87178291200.0);        // -1/14!
    c = fma(c, z, 1.0 / 479001600.0);           //  1/12!
    c = fma(c, z, -1.0 / 3628800.0);            // -1/10!
    c = fma(c, z, 1.0 / 40320.0);               //  1/8!
    c = fma(c, z, -1.0 / 720.0);                // -1/6!
    c = fma(c, z, 1.0 / 24.0);                  //  1/4!
    c = fma(c, z, -0.5);
    cs = fma(c, z, 1.0);
}

// Orders LDS writes before later LDS reads of the same wave (no s_barrier:
// the caller is a single wave, possibly while other waves of the block run on).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Value of `v` in lane `lane` (a constant), the same in every lane afterwards.
__device__ __forceinline__ double lane_bcast(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// The loads behind the running totals, issued by request_sums() and consumed by
// solve_and_update().  Vector loads return in issue order, so the wave that solves asks
// for these before anything else it loads.
//
// The running totals and their per-iteration changes are kept in 64-bit FIXED POINT: sum t is
// stored as round(value * 2^fx[t]).  Integer addition is associative, so the totals do not
// depend on the order in which the tiles' atomic additions (or the ranks of a sharded run)
// arrive: two runs on the same input give the same bits, and every rank of a sharded run
// solves from the same bits.  fx[t] is chosen on the host from bounds of the sums (number of
// points, largest coordinate, largest normal component, threshold) so that the total stays
// below 2^62; the resolution is finer than a double's for every sum.
typedef long long fx_t;

struct SumsInFlight {
    fx_t base, delta[kDeltaCopies];
    int exponent;
};

__device__ __forceinline__ SumsInFlight request_sums(const fx_t *base, const fx_t *delta, const int *fx)
{
    const int t = threadIdx.x & 63;
    SumsInFlight s;
    s.base = 0;
    s.exponent = 0;
#pragma unroll
    for (int cp = 0; cp < kDeltaCopies; ++cp) s.delta[cp] = 0;
    if (t < 30) {
        s.base = base[t];
#pragma unroll
        for (int cp = 0; cp < kDeltaCopies; ++cp) s.delta[cp] = delta[cp * kNumSums + t];
        s.exponent = fx[t];
    }
    return s;
}

// value -> fixed point (round to nearest)
__device__ __forceinline__ fx_t to_fixed(double v, int exponent)
{
    return __double2ll_rn(ldexp(v, exponent));
}

// Wave-parallel solve, called by ONE full wave (64 lanes) with the sums requested by
// request_sums().  Results: R (9) and T (3) to out_rt (LDS); when `record`, also the sums
// to totals[], the state, the accumulated transform and the history rows.
//
// Gaussian elimination of the diagonally scaled, positive definite matrix.  Lane r
// (r < 6) keeps row r and element r of the right-hand side in registers; what other rows
// need of it travels by lane broadcasts, so there is no LDS or memory round trip inside
// the elimination.  Lanes >= 6 run the same instructions on values nobody reads.
// Restates CPoint2PlaneICP.cpp:109 (X), :111-119 (R = Rz*Ry*Rx, T) and
// :36-39 / :52-55 (Trans = [R T; 0 1] * Trans).
__device__ inline void solve_and_update(const SumsInFlight &sums, fx_t *totals,
                                        IcpState *st, double *X_hist, double *info_hist,
                                        int iter, bool record, double *out_rt,
                                        unsigned long long *trace_row = nullptr)
{
    const int t = threadIdx.x & 63;
    bool ok = true;

    // running totals of the normal equations: the totals after the iteration
    // before, plus what the tiles that changed since have added or taken away.
    // Upper triangle: element (r, c), r <= c, sits at r*6 - r(r-1)/2 + (c - r).
    double stat = 0.0;                      // lanes 0..29 carry sum t
    fx_t fixed = 0;
    if (t < 30) {
        fixed = sums.base;
#pragma unroll
        for (int cp = 0; cp < kDeltaCopies; ++cp) fixed += sums.delta[cp];
        stat = ldexp((double)fixed, -sums.exponent);
    }
    if (record && t < kNumSums) totals[t] = fixed;
    SVICP_SOLVE_STAMP(10);
    // lane r < 6 takes row r of the matrix and element r of the right-hand side
    const int rr6 = t < 6 ? t : 5;
    double m[6];
    // which lane holds element (rr6, c): the triangle index lo*6 - lo(lo-1)/2 + (hi - lo) of
    // rows 0..5, five bits each, one constant per column (a shift and a mask per lane instead
    // of the index arithmetic)
    constexpr auto tri = [](int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r;
                                            return lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo); };
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const unsigned packed = (unsigned)tri(0, c) | (unsigned)tri(1, c) << 5 | (unsigned)tri(2, c) << 10 |
                                (unsigned)tri(3, c) << 15 | (unsigned)tri(4, c) << 20 | (unsigned)tri(5, c) << 25;
        m[c] = __shfl(stat, (int)((packed >> (5 * rr6)) & 31u), 64);
    }
    double rhs = __shfl(stat, 21 + rr6, 64);

    // scaling d ~ diag^-1/2, a power of two (exact)
    double d[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double a = lane_bcast(m[c], c);
        const bool good = a > 1e-300 && a < 1e300;
        ok = ok && good;
        d[c] = good ? pow2_rsqrt(a) : 1.0;
    }
    double dmine = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) dmine = (rr6 == c) ? d[c] : dmine;
#pragma unroll
    for (int c = 0; c < 6; ++c) m[c] = m[c] * dmine * d[c];
    rhs *= dmine;

    SVICP_SOLVE_STAMP(11);
    // Elimination without a division on the critical path.  Step c replaces row r > c by
    // sigma * (p * row_r - a_rc * row_c), p the pivot: for any sigma > 0 that is the
    // elimination step followed by a scaling of the row, which changes neither the
    // solution nor the stability (the matrix is positive definite, no pivoting needed).
    // sigma is the hardware's estimate of 1 / p, so the rows keep their magnitude; the
    // dependent chain per step is estimate, two products and one fused multiply-add
    // instead of a square root refined to full precision.
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double p = lane_bcast(m[c], c);
        if (!(p > 1e-14) || !(p < 1e300)) { ok = false; p = 1.0; }
        const double sigma = __builtin_amdgcn_rcp(p);
        const bool below = rr6 > c;
        const double ps = below ? p * sigma : 1.0;          // rows <= c stay as they are
        const double ms = below ? m[c] * sigma : 0.0;
#pragma unroll
        for (int k = c + 1; k < 6; ++k) m[k] = fma(-ms, lane_bcast(m[k], c), ps * m[k]);
        rhs = fma(-ms, lane_bcast(rhs, c), ps * rhs);
    }
    SVICP_SOLVE_STAMP(12);
    // lane r now holds row r of an upper triangular system; 1 / diagonal to full
    // precision in every lane at once, then the substitution from the last row up
    double diag = m[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) diag = (rr6 == c) ? m[c] : diag;
    if (!(fabs(diag) > 1e-300)) diag = 1.0;                  // (ok is already false then)
    double inv = __builtin_amdgcn_rcp(diag);
#pragma unroll
    for (int k = 0; k < 2; ++k) inv = fma(inv, fma(-diag, inv, 1.0), inv);
    double x[6];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        x[r] = lane_bcast(rhs * inv, r);
        if (r > 0) rhs = fma(-m[r], x[r], rhs);             // rows above take x[r] out
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        x[r] *= d[r];
        if (!(fabs(x[r]) < 1e300)) ok = false;
    }
    if (!ok) {
#pragma unroll
        for (int r = 0; r < 6; ++r) x[r] = 0.0;
    }
    SVICP_SOLVE_STAMP(13);
    // the three angles' sines and cosines on three lanes at once
    double sn, cs;
    const double ang = t == 0 ? x[0] : t == 1 ? x[1] : x[2];
    if (__builtin_expect(fabs(x[0]) <= 0.5 && fabs(x[1]) <= 0.5 && fabs(x[2]) <= 0.5, 1))
        sincos_small(ang, sn, cs);
    else
        sincos(ang, &sn, &cs);       // (long: kept out of the usual path)
    const double sx = lane_bcast(sn, 0), cx = lane_bcast(cs, 0), sy = lane_bcast(sn, 1),
                 cy = lane_bcast(cs, 1), sz = lane_bcast(sn, 2), cz = lane_bcast(cs, 2);
    // R = Rz(X2) * Ry(X1) * Rx(X0)
    const double R[9] = { cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                          sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                          -sy,     cy * sx,                cy * cx };
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) out_rt[k] = R[k];
        out_rt[9] = x[3]; out_rt[10] = x[4]; out_rt[11] = x[5];
    }
    SVICP_SOLVE_STAMP(14);
    if (__builtin_expect(!record, 1)) return;

    // Trans = [R T; 0 1] * Trans, one element per lane
    double tnew = 0.0;
    if (t < 12) {
        const int r = t >> 2, c = t & 3;
        const double r0 = r == 0 ? R[0] : r == 1 ? R[3] : R[6];
        const double r1 = r == 0 ? R[1] : r == 1 ? R[4] : R[7];
        const double r2 = r == 0 ? R[2] : r == 1 ? R[5] : R[8];
        // the first iteration of a run starts from the identity (CPoint2PlaneICP.cpp:33)
        const double p0 = iter == 0 ? (c == 0 ? 1.0 : 0.0) : st->Trans[c];
        const double p1 = iter == 0 ? (c == 1 ? 1.0 : 0.0) : st->Trans[4 + c];
        const double p2 = iter == 0 ? (c == 2 ? 1.0 : 0.0) : st->Trans[8 + c];
        tnew = r0 * p0 + r1 * p1 + r2 * p2;
        if (c == 3) tnew += (r == 0 ? x[3] : r == 1 ? x[4] : x[5]);
    }
    if (iter == 0 && t >= 12 && t < 16) tnew = t == 15 ? 1.0 : 0.0;   // last row of the 4x4
    wave_lds_sync();
    if (t < 12 || (iter == 0 && t < 16)) st->Trans[t] = tnew;
    if (t < kNumSums) st->S[t] = stat;
    const double s27 = lane_bcast(stat, 27), s28 = lane_bcast(stat, 28), s29 = lane_bcast(stat, 29);
    if (t != 0) return;

    if (iter == 0) st->status = ok ? 0 : 4;   // SVICP_ERR_SINGULAR
    else if (!ok) st->status = 4;
#pragma unroll
    for (int k = 0; k < 9; ++k) st->R[k] = R[k];
    st->T[0] = x[3]; st->T[1] = x[4]; st->T[2] = x[5];
    const double step = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3] + x[4] * x[4] +
                             x[5] * x[5]);
    st->S[30] = step;
    if (X_hist) {
#pragma unroll
        for (int r = 0; r < 6; ++r) X_hist[6 * iter + r] = x[r];
    }
    if (info_hist) {
        info_hist[4 * iter + 0] = s27;
        info_hist[4 * iter + 1] = s28;
        info_hist[4 * iter + 2] = step;
        info_hist[4 * iter + 3] = s29;
    }
}

// ------------------------------------------------------------------------
// The iteration kernel: one launch per ICP iteration, one 1024-point tile of
// the source per 256-thread block.
//
//  0. (APPLY) the first wave solves the previous iteration's 6x6 system from
//     its 32 sums and forms R, T (CPoint2PlaneICP.cpp:109-119); every block does
//     this redundantly, so no separate solve launch and no broadcast is needed.
//     The block of tile 0 also records X and accumulates the 4x4 transform
//     (CPoint2PlaneICP.cpp:36-39, :52-55).
//  1. Every point is updated (CPoint2PlaneICP.cpp:42-50: fp64 math, one rounding
//     to fp32).  With CERTIFY the kernel then tries to prove that the point's
//     correspondence of the previous iteration still holds: a point that moved
//     by delta has every other target point at least lb - delta away (triangle
//     inequality); if its distance to the remembered neighbour is smaller than
//     that, the neighbour is still the unique nearest one and no search is
//     needed.  The bounds carry 1e-5 of relative slack, far more than fp32
//     rounding of the distances, so the correspondence is the one a full search
//     returns, bit for bit.  Proven points are accumulated from the remembered
//     target point and normal: pure streaming, no gather.
//  2. The remaining points are queued in LDS in a fixed order and searched by
//     the whole block (CPoint2PlaneICP.cpp:65-67); what the search finds is
//     remembered for the next iteration.
//  3. Block sums of the 30 accumulators replace row `tile` of the partials and
//     their change is added to this iteration's delta sums, so no separate
//     reduction launch is needed; a tile that was skipped adds nothing.
// ------------------------------------------------------------------------
// Blocks that share an XCD (blockIdx % 8 equal) take neighbouring work items,
// so each XCD's L2 sees one contiguous stretch of the Morton-ordered source.
__device__ __forceinline__ int xcd_contiguous(int block, int nb)
{
    const int q8 = nb >> 3, r8 = nb & 7;
    const int xcd = block & 7, within = block >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
}

// Inclusive prefix sum over the 64 lanes by data-parallel-primitive moves.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ int dpp_get(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, BOUND);
}
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += dpp_get<0x111, 0xf, true>(v);    // row_shr:1 (0 from outside the row of 16)
    v += dpp_get<0x112, 0xf, true>(v);    // row_shr:2
    v += dpp_get<0x114, 0xf, true>(v);    // row_shr:4
    v += dpp_get<0x118, 0xf, true>(v);    // row_shr:8   -> scan inside each row
    v += dpp_get<0x142, 0xa, false>(v);   // row_bcast:15: rows 1 and 3 add the row before
    v += dpp_get<0x143, 0xc, false>(v);   // row_bcast:31: rows 2 and 3 add rows 0 + 1
    return v;
}

// Which tile block `b` takes in iteration `iter`, given which tiles had work in the
// iteration before (`prev`, one byte per tile).  Once a registration has settled, few
// tiles still have work (points that keep changing in their last fp32 bits) and they are
// the same from one iteration to the next, but they come in runs with power-of-two
// strides (they follow the coordinate planes through the Morton order), which a fixed
// block order keeps stacking on the same compute units: the launch then ends with the
// unit that got three of them.  The dispatcher places block b of a fresh launch by
// b mod 256 (XCD b mod 8, compute unit (b / 8) mod 32), so here the tiles that had work
// take the first blocks, one per compute unit while they are at most 256, neighbours on
// one XCD, and the tiles without work take the rest.  With every tile active this is
// xcd_contiguous itself.  Every wave of every block evaluates the same flags, so the
// result is a permutation of the tiles.  `predicted`: the tile had work last time.
// Position of the r-th (from 0) set bit of the 64-bit mask m; r < popcount(m).
__device__ __forceinline__ int select64(unsigned long long m, int r)
{
    int pos = 0;
    int c = __popc((unsigned)m);
    if (r >= c) { r -= c; pos += 32; m >>= 32; }
    unsigned w = (unsigned)m;
    c = __popc(w & 0xffffu);
    if (r >= c) { r -= c; pos += 16; w >>= 16; }
    c = __popc(w & 0xffu);
    if (r >= c) { r -= c; pos += 8; w >>= 8; }
    c = __popc(w & 0xfu);
    if (r >= c) { r -= c; pos += 4; w >>= 4; }
    c = __popc(w & 0x3u);
    if (r >= c) { r -= c; pos += 2; w >>= 2; }
    if (r >= (int)(w & 1u)) pos += 1;
    return pos;
}

__device__ __forceinline__ int choose_tile(const unsigned long long bits, int nb, int b, bool &predicted,
                                           int &n_predicted)
{
    const int lane = threadIdx.x & 63;                             // bits: the flags of tiles 64 lane .. 64 lane + 63
    const int valid = min(max(nb - 64 * lane, 0), 64);             // tiles past the end: neither kind
    const unsigned long long in_range = valid >= 64 ? ~0ull : (1ull << valid) - 1ull;
    const unsigned long long with_work = bits & in_range, without = ~bits & in_range;
    const int cnt = __popcll(with_work);
    const int inc = wave_inclusive_scan(cnt);
    const int A = __builtin_amdgcn_readlane(inc, 63);
    n_predicted = A;
    predicted = b < A;
    // the j-th tile with (predicted) or without work: lane L holds it
    const int j = predicted ? xcd_contiguous(b, A) : b - A;
    const int mine = predicted ? inc : min(64 * (lane + 1), nb) - inc;   // inclusive count of my kind
    const unsigned long long kind = predicted ? with_work : without;
    const int pos = select64(kind, max(j - (mine - (int)__popcll(kind)), 0));   // meaningful in lane L only
    const unsigned long long m = __ballot(mine > j);
    if (m == 0ull) return xcd_contiguous(b, nb);                   // cannot happen: j < count
    const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
    const int tile = 64 * L + __builtin_amdgcn_readlane(pos, L);
    return tile < nb ? tile : xcd_contiguous(b, nb);
}

#ifndef SVICP_TILE_OCC
#define SVICP_TILE_OCC 4
#endif
#ifndef SVICP_MANY_OCC
#define SVICP_MANY_OCC 3
#endif
// The block of tile 0 clears what the next iteration adds to: its delta row (the buffer holds
// one row more than the longest run; row 0 is cleared by the kernel that ends a run) and the
// flag row it will write (last read in the iteration before this one).
__device__ __forceinline__ void start_next_rows(fx_t *__restrict__ deltas, unsigned long long *active,
                                                int flag_words, int iter, int tile)
{
    if (tile != 0) return;
    for (int k = threadIdx.x; k < kDeltaRow; k += kItThreads) deltas[(size_t)(iter + 1) * kDeltaRow + k] = 0;
    if (active)
        for (int w = threadIdx.x; w < flag_words; w += kItThreads)
            active[(size_t)((iter + 1) % kFlagRows) * flag_words + w] = 0ull;
}

// The work of one block: tile `tile` of a source of `nb` tiles.
// INPLACE: the coordinates are updated where they are read (s?_in are not looked at); else
// s?_in and s?_out may still be the same arrays (the batch decides per launch).
// solve_now = false: a kernel launched before this one has solved the previous iteration's
// system and left R (9) and T (3) at rt_given (the batch does that: one solve per pair instead
// of one per block).
// BOXES: after the update every wave leaves the box of its points in ps.box (see box_cannot_move),
// and the routine is ready to be called in a loop.
template <bool APPLY, bool CERTIFY, int CH, bool INPLACE = false, bool BOXES = false>
__device__ __forceinline__ void iterate_tile(
    const GridDesc &g, const float *sx_in_, const float *sy_in_, const float *sz_in_,
    float *sx_out, float *sy_out, float *sz_out, int n, fx_t *__restrict__ totals,
    fx_t *__restrict__ deltas, const int *__restrict__ fx,
    IcpState *st, double *X_hist, double *info_hist, int iter, const CohParams &cp,
    const PointState &ps, double *__restrict__ partials, const int fixed_tile, const int nb,
    const bool solve_now = true, const double *__restrict__ rt_given = nullptr)
{
    constexpr int NP = kItPoints * kItHalves;         // points per thread
    __shared__ float4 queue[kItTile];                 // {x, y, z, index in tile}
    __shared__ double red[kItThreads / 64][kNumSums];
    __shared__ int wcount[NP][kItThreads / 64];
    __shared__ double shRT[12];
    __shared__ float4 found[kItThreads / 64][2];
    __shared__ __attribute__((aligned(16))) int wave_moved[kItThreads / 64];

    // BOXES: the routine runs in a loop over tiles.  It has no vector register to spare, and
    // whatever it derives from the thread's number alone is the same for every tile: the
    // compiler would work that out once, before the loop, and then has to park it in scratch
    // memory (43 registers' worth, each reload a wait for memory in the middle of the routine).
    // So the number passes through an empty asm statement per tile: nothing derived from it can
    // be carried round the loop.
    unsigned tid = threadIdx.x;
    if (BOXES) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wv = tid >> 6;
    const float *const sx_in = INPLACE ? sx_out : sx_in_, *const sy_in = INPLACE ? sy_out : sy_in_,
                *const sz_in = INPLACE ? sz_out : sz_in_;
    const bool in_place = INPLACE || sx_in == sx_out;
#ifdef SVICP_TRACE
    const unsigned long long t_entry = wall_clock64();
#endif

    // ---- the wave that solves asks for the sums before any other load ----
    SumsInFlight sums;
    if (APPLY && wv == 0 && solve_now)
        sums = request_sums(totals + (size_t)(iter - 1) * kNumSums,
                            deltas + (size_t)(iter - 1) * kDeltaRow, fx);
    // ---- which tile: fixed_tile, or (fixed_tile < 0) block blockIdx.x's place in the
    // order that puts the tiles with work first (choose_tile) ----
    int tile = fixed_tile;
    bool prefetch = false;
    if (APPLY && CERTIFY && fixed_tile < 0) {
        bool predicted;
        int n_predicted;
        tile = choose_tile(ps.active[((iter - 1) % kFlagRows) * kFlagWords + lane], nb, (int)blockIdx.x,
                           predicted, n_predicted);
        // while most tiles still have work the points move too far for the remembered
        // correspondences to be of use
        // (only where this block's first wave solves: the early loads sit in the other arm of
        // that branch.  With the solve in a launch of its own there is no such arm, and a
        // block that took `prefetch` for done would go on with state it never loaded.)
        prefetch = solve_now && predicted && 4 * n_predicted <= nb;
    }
    // Behind the flag rows: per tile the position of the last point of the tile that had to be
    // searched once the registration had settled.  Such points are near-ties that come back
    // launch after launch; the last wave asks for the lines their search reads while the first
    // wave solves, so that the search, which ends the launch, finds them in the cache.
    float4 *const hints = reinterpret_cast<float4 *>(ps.active + kFlagRows * kFlagWords);
    float4 hint = make_float4(__int_as_float(0x7fc00000), 0.f, 0.f, 0.f);
    if (APPLY && CERTIFY && prefetch && wv == kItThreads / 64 - 1) hint = hints[tile];
    const long long tile0 = (long long)tile * kItTile;
    // everything per point is addressed as (array + tile0)[small index]: a uniform base
    // and a 32-bit offset per thread
    const int tile_n = (int)min((long long)kItTile, (long long)n - tile0);   // points in this tile
    const float *const bx_in = sx_in + tile0, *const by_in = sy_in + tile0, *const bz_in = sz_in + tile0;
    float *const bx_out = sx_out + tile0, *const by_out = sy_out + tile0, *const bz_out = sz_out + tile0;
    Float3 *const btp = ps.tp + tile0, *const bnv = ps.nv + tile0;
    float *const blb = ps.lb + tile0;

#ifdef SVICP_TRACE
    if (tid == 0 && iter < kTraceIters && tile < kTraceTiles) {
        g_trace[((size_t)iter * kTraceTiles + tile) * kTraceStamps + 0] = t_entry;
        g_trace[((size_t)iter * kTraceTiles + tile) * kTraceStamps + 8] =
            __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        g_trace[((size_t)iter * kTraceTiles + tile) * kTraceStamps + 9] =
            __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
    float4 *const boxes = BOXES && APPLY && ps.box ? ps.box + (size_t)tile * kBoxRecord : nullptr;
    // ---- A: coordinates (and the tile's old sums) are requested first, so
    // their latency runs under the solve ----
    float px[NP], py[NP], pz[NP];
    bool live[NP];
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) {
        const unsigned li = sl * kItThreads + tid;
        live[sl] = (int)li < tile_n;
        const unsigned lc = live[sl] ? li : 0u;
        px[sl] = bx_in[lc]; py[sl] = by_in[lc]; pz[sl] = bz_in[lc];
    }
    start_next_rows(deltas, ps.active, ps.flag_words, iter, tile);

    // ---- B: the first wave solves the previous iteration's system.  Meanwhile, in a tile
    // that is expected to have work (it had some in the iteration before), the other waves
    // ask for their points' remembered correspondences, and their first lanes touch the
    // lines the first wave will want, so that its own loads after the solve find them in
    // the cache.  (Two arms of one branch: the solve's registers and the early loads'
    // registers are never alive together.) ----
    Float3 tp[NP], nv[NP];
    float lb[NP];
    float touched = 0.f, touched_too = 0.f;
    double R0 = 1, R1 = 0, R2 = 0, R3 = 0, R4 = 1, R5 = 0, R6 = 0, R7 = 0, R8 = 1, T0 = 0, T1 = 0,
           T2 = 0;
    if (APPLY && !solve_now) {
        R0 = rt_given[0]; R1 = rt_given[1]; R2 = rt_given[2]; R3 = rt_given[3]; R4 = rt_given[4];
        R5 = rt_given[5]; R6 = rt_given[6]; R7 = rt_given[7]; R8 = rt_given[8]; T0 = rt_given[9];
        T1 = rt_given[10]; T2 = rt_given[11];
    } else if (APPLY) {
        // totals[k]: sums after iteration k - 1 (row 0 is zero); deltas[k]: changes made by iteration k
        if (wv == 0) {
            solve_and_update(sums, totals + (size_t)iter * kNumSums, st, X_hist, info_hist, iter - 1,
                             // Every block solves the same system; ONE records the result.  The
                             // launch's last block does: once a registration has settled it holds a
                             // tile without work (choose_tile gives the first blocks to the tiles
                             // with work), so the record's trip to memory for the accumulated
                             // transform delays nobody -- in the block of tile 0, which usually has
                             // work, it ended the launch whenever no tile had to search.
                             blockIdx.x + 1 == gridDim.x, shRT
#ifdef SVICP_TRACE
                             , (iter < kTraceIters && tile < kTraceTiles)
                                   ? g_trace + ((size_t)iter * kTraceTiles + tile) * kTraceStamps : nullptr
#endif
                             );
        } else if (CERTIFY && prefetch) {
#pragma unroll
            for (int sl = 0; sl < NP; ++sl) {
                const unsigned li = sl * kItThreads + tid;
                const unsigned lc = live[sl] ? li : 0u;
                tp[sl] = btp[lc]; nv[sl] = bnv[lc]; lb[sl] = blb[lc];
            }
            // the first wave's share: per slot 64 points = 6 + 6 + 2 lines of 128 bytes (a few
            // more where the 768-byte stretches straddle lines: one touch every 96 bytes)
            const int u = (int)tid - 64;
            if (u < NP * 18) {
                const int sl = u / 18, line = u % 18;
                // a point every 8 records of 12 bytes, every 32 of 4; kept inside the tile at
                // the end of the source
                const unsigned pt = min((unsigned)(sl * kItThreads + (line < 16 ? 8 * (line & 7) : 32 * (line - 16))),
                                        (unsigned)(tile_n - 1));
                touched = line < 8 ? btp[pt].x : line < 16 ? bnv[pt].x : blb[pt];
            } else if (u < NP * 18 + 2) {
                // and the two lines of the tile's old sums, which the end of the block asks for
                touched = (float)partials[(size_t)tile * kNumSums + 16 * (u - NP * 18)];
            }
            if (wv == kItThreads / 64 - 1) probe_touch(g, hint.x, hint.y, hint.z, touched, touched_too);
        }
        __syncthreads();
        R0 = shRT[0]; R1 = shRT[1]; R2 = shRT[2]; R3 = shRT[3]; R4 = shRT[4]; R5 = shRT[5];
        R6 = shRT[6]; R7 = shRT[7]; R8 = shRT[8]; T0 = shRT[9]; T1 = shRT[10]; T2 = shRT[11];
    }
    asm volatile("" ::"v"(touched), "v"(touched_too));   // the touching loads are issued whatever the optimiser thinks of them
    SVICP_STAMP(1);

    // ---- C: update (CPoint2PlaneICP.cpp:44-49) ----
    float m2[NP];                 // squared displacement of the point
    bool any_moved = false;
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) {
        m2[sl] = 0.f;
        if (APPLY) {
            const float ox = px[sl], oy = py[sl], oz = pz[sl];
            // ((R0*x + R1*y) + R2*z) + T0 in fp64, no contraction, one rounding to fp32
            double dx = (double)ox, dy = (double)oy, dz = (double)oz;
            px[sl] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R0, dx), __dmul_rn(R1, dy)), __dmul_rn(R2, dz)), T0);
            py[sl] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R3, dx), __dmul_rn(R4, dy)), __dmul_rn(R5, dz)), T1);
            pz[sl] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R6, dx), __dmul_rn(R7, dy)), __dmul_rn(R8, dz)), T2);
            const bool moved = px[sl] != ox || py[sl] != oy || pz[sl] != oz;
            any_moved = any_moved || (live[sl] && moved);
            float mx = px[sl] - ox, my = py[sl] - oy, mz = pz[sl] - oz;
            m2[sl] = moved ? mx * mx + my * my + mz * mz : 0.f;
            // in place the store can be skipped for a point that did not move
            if (live[sl] && (moved || !in_place) && !((kDrop & kDropSrcWrite) && iter == kDropIter)) {
                const unsigned li = sl * kItThreads + tid;
                bx_out[li] = px[sl]; by_out[li] = py[sl]; bz_out[li] = pz[sl];
            }
        }
    }

    // Where this wave's points lie now, for the next launches' proof that the tile stays put --
    // but only once the wave has all but come to rest (no point moved by more than 2^-20 of its
    // largest coordinate, some eight units in the last place): the steps shrink by a few per cent
    // per iteration, so a wave that still moves farther will not be static in the next launch,
    // and the launches that search, which are bound by their arithmetic, do not pay for six
    // reductions per wave.  (A coordinate that is not finite fails the comparison as well.)
    if (BOXES && APPLY && boxes) {
        // (the first point of every lane first: in a launch that searches, that settles it)
        bool at_rest = true;
#pragma unroll
        for (int sl = 0; sl < NP; ++sl) {
            if (sl == 1 && __ballot(!at_rest) != 0ull) break;
            const float big = fmaxf(fmaxf(fabsf(px[sl]), fabsf(py[sl])), fabsf(pz[sl])) * 0x1p-20f;
            at_rest = at_rest && (!live[sl] || m2[sl] <= big * big);
        }
        float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
        if (__ballot(!at_rest) == 0ull) {
#pragma unroll
            for (int sl = 0; sl < NP; ++sl)
                if (live[sl]) {
                    lox = fminf(lox, px[sl]); loy = fminf(loy, py[sl]); loz = fminf(loz, pz[sl]);
                    hix = fmaxf(hix, px[sl]); hiy = fmaxf(hiy, py[sl]); hiz = fmaxf(hiz, pz[sl]);
                }
            lox = wave_min_f(lox); loy = wave_min_f(loy); loz = wave_min_f(loz);
            hix = wave_max_f(hix); hiy = wave_max_f(hiy); hiz = wave_max_f(hiz);
        }
        if (lane == 0) {
            boxes[wv * kBoxWords] = make_float4(lox, loy, loz, 0.f);
            boxes[wv * kBoxWords + 1] = make_float4(hix, hiy, hiz, 0.f);
        }
    }

    // ---- D: a tile none of whose points moves keeps its sums.  A point whose
    // update rounds to the same three floats is where it was in the previous
    // iteration, so its nearest neighbour, weight and row are the same bits; if
    // that holds for the whole tile, its row of `partials` is still right and
    // only the coordinates had to be read.  Near convergence most tiles are
    // like that. ----
    SVICP_STAMP(2);
    if (APPLY && CERTIFY && in_place) {
        // one barrier: every wave leaves its own verdict in its own word (nothing to clear
        // beforehand), and every thread reads the four words at once
        static_assert(kItThreads / 64 == 4, "the four verdicts are read as one 16-byte word");
        const int wave_verdict = __ballot(any_moved) != 0ull ? 1 : 0;       // (every lane votes)
        if (lane == 0) wave_moved[wv] = wave_verdict;
        __syncthreads();
        const int4 verdicts = *reinterpret_cast<const int4 *>(wave_moved);
        if (!(verdicts.x | verdicts.y | verdicts.z | verdicts.w)) return;   // (its flag bit stays clear)
    }
    // (the tile's flag bit is raised at the very end: an atomic stays in the wave's count of
    // outstanding memory operations for microseconds, and every later load would wait behind it)
    SVICP_STAMP(3);

    double acc[30];
#pragma unroll
    for (int v = 0; v < 30; ++v) acc[v] = 0.0;

    // ---- E/F: try to keep the previous correspondences ----
    bool need_search[NP];
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) need_search[sl] = live[sl];
    if (CERTIFY) {
        // lb never exceeds cov, so a point that moved farther than that cannot
        // be proven; when that is the whole wave the state is not even read
        bool hopeless = true;
#pragma unroll
        for (int sl = 0; sl < NP; ++sl) hopeless = hopeless && !(live[sl] && m2[sl] < cp.cov * cp.cov);
        if (kLateState && iter <= kLateState) hopeless = true;
        if (__ballot(!hopeless) != 0ull) {
            if (!prefetch || wv == 0) {
#pragma unroll
                for (int sl = 0; sl < NP; ++sl) {
                    const unsigned li = sl * kItThreads + tid;
                    const unsigned lc = live[sl] ? li : 0u;
                    tp[sl] = btp[lc]; nv[sl] = bnv[lc]; lb[sl] = blb[lc];
                }
            }
#pragma unroll
            for (int sl = 0; sl < NP; ++sl) {
                const unsigned li = sl * kItThreads + tid;
                float lbn = lb[sl];
                if (m2[sl] > 0.f) lbn = bound_after_move(lbn, m2[sl]);
                const bool has = tp[sl].x == tp[sl].x;          // a neighbour is remembered
                float dx = px[sl] - tp[sl].x, dy = py[sl] - tp[sl].y, dz = pz[sl] - tp[sl].z;
                float d2 = __fmul_rn(dx, dx);
                d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
                d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
                // sqrt(d2) * (1 + 1e-5) < lbn, compared as squares
                // a point that did not move keeps what the last search or proof
                // at this very position established, near-ties included
                const bool still = m2[sl] == 0.f;
                const bool proven_nn =
                    has && (still || (lbn > 0.f && d2 * kSlackUp2 < lbn * lbn));
                const bool proven_none = !has && (still || lbn > cp.root_hi);
                need_search[sl] = live[sl] && !(proven_nn || proven_none);
                if (live[sl] && !need_search[sl] && m2[sl] > 0.f) blb[li] = lbn;
                if (live[sl] && proven_nn && d2 <= cp.thr)
                    accumulate_point(acc, px[sl], py[sl], pz[sl], d2, cp.thr, tp[sl].x, tp[sl].y,
                                     tp[sl].z, nv[sl].x, nv[sl].y, nv[sl].z);
            }
        }
    }

    SVICP_STAMP(4);
    // ---- queue of the points that need a search, in a fixed order: by slot,
    // then by thread (ballot + popcount per wave, wave totals through LDS) ----
    int my_rank[NP];
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) {
        const unsigned long long mask = __ballot(need_search[sl]);
        my_rank[sl] = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[sl][wv] = __popcll(mask);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) {
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kItThreads / 64; ++w) {
            const int c = wcount[sl][w];
            before += w < wv ? c : 0;
            all += c;
        }
        if (need_search[sl]) {
            // the thread's index formed anew from the lane counter: kept from the top of the
            // kernel it would be one more value to hold in registers across every phase
            const int me = 64 * wv + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            queue[total + before + my_rank[sl]] =
                make_float4(px[sl], py[sl], pz[sl], __int_as_float(sl * kItThreads + me));
        }
        total += all;
    }
    if (total > 0) __syncthreads();     // (the same in every thread)


    SVICP_STAMP(5);
    // ---- G: search the queued points (CPoint2PlaneICP.cpp:65-67) ----
    // what a search found is remembered and, inside the threshold, accumulated
    const bool dropping = kDrop != 0 && iter == kDropIter;
    auto settle = [&](const float4 &e, float best, int bpos, float second, auto &&target_point) {
        const unsigned li = (unsigned)__float_as_int(e.w);
        Float3 tp = { __int_as_float(0x7fc00000), 0.f, 0.f };   // NaN: no neighbour
        Float3 nv = { 0.f, 0.f, 0.f };
        // a point without an accepted neighbour has every target farther than cov / kSlackUp
        // (computed), hence farther than cov * kSlackDown^2 in truth
        float lbn = cp.cov * kSlackDown * kSlackDown;
        if (bpos >= 0 && sqrtf(best) * kSlackUp <= cp.cov) {
            // no unprobed point can be nearer: bpos is the nearest neighbour
            const float4 t = ((kDrop & (kDropPts | kDropQuad | kDropDir)) && dropping)
                                 ? make_float4(e.x + 0.5f, e.y, e.z, 0.f) : target_point();
            tp = { t.x, t.y, t.z };
            if ((kDrop & kDropNrm) && dropping) nv = { 0.f, 0.6f, 0.8f };
            else nv = reinterpret_cast<const Float3 *>(g.nrm)[bpos];     // one 12-byte gather, not three of 4
            lbn = fminf(sqrtf(second) * kSlackDown, lbn);
            if (best <= cp.thr)
                accumulate_point(acc, e.x, e.y, e.z, best, cp.thr, t.x, t.y, t.z, nv.x, nv.y, nv.z);
        }
        if ((kDrop & kDropStateWrite) && dropping) return;
        if (kLateState && iter < kLateState) return;
        btp[li] = tp;
        bnv[li] = nv;
        blb[li] = lbn;
    };
    // a handful of points (the usual case once the registration has settled): one wave
    // per point, the candidates spread over its lanes; the answers wait in LDS
    const bool few = total <= kItThreads / 64;
    if (few && total > 0) {
        if (wv < total) {
            const float4 e = queue[wv];
            float best, second; int bpos, bidx;
            float4 t;
            nn_capped_wave<true>(g, e.x, e.y, e.z, best, bpos, bidx, t, &second);
            if (lane == 0) {
                found[wv][0] = make_float4(t.x, t.y, t.z, best);
                found[wv][1] = make_float4(second, __int_as_float(bpos), 0.f, 0.f);
            }
        }
        __syncthreads();
    }
    for (int q = tid; q < total; q += kItThreads) {
        const float4 e = queue[q];
        float best, second; int bpos, bidx;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        // with the proof in use the searching case tolerates being the far branch (without
        // it, it is the only case, and laid out as the far one it spills)
        if (CERTIFY ? __builtin_expect(few, 1) : few) {
            const float4 f0 = found[q][0], f1 = found[q][1];
            t = f0; best = f0.w; second = f1.x; bpos = __float_as_int(f1.y);
        } else {
            nn_capped<CH, true>(g, e.x, e.y, e.z, best, bpos, bidx, &second, dropping);
        }
        settle(e, best, bpos, second, [&]() { return few ? t : g.pts[bpos]; });
    }

    SVICP_STAMP(6);
    // ---- H: the tile's new sums replace its row; the difference goes to this
    // iteration's delta (one fp64 atomic per changed sum) ----
    // the tile's old sums are asked for here, late: nothing else has to stay in registers
    // for them, and the block sum covers their latency
    double old_row = 0.0;
    int exponent = 0;
    // (the threads that need the scaling are the lanes that asked for it with the sums)
    if (tid < 30) exponent = (APPLY && solve_now) ? sums.exponent : fx[tid];
    // (a tile's 32 sums lie together: one request of two lines)
    if (APPLY && tid < 30) old_row = partials[(size_t)tile * kNumSums + tid];
    const double s = block_sum_30(acc, red);
    if (tid < 30) {
        partials[(size_t)tile * kNumSums + tid] = s;
        // both rows go to fixed point before they are subtracted: the totals are then the exact
        // integer sum of the tiles' current rows, whatever the order of the additions
        const fx_t d = to_fixed(s, exponent) - to_fixed(old_row, exponent);
        if (d != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(deltas) + (size_t)iter * kDeltaRow +
                          (tile & (kDeltaCopies - 1)) * kNumSums + tid,
                      (unsigned long long)d);
    }
    // slot 31 of the iteration's delta row counts the searched points (a statistic for the
    // caller; the solve reads slots 0..29 only).  Like every atomic of this kernel it is issued
    // after the wave's last load.
    if (tid == 31 && total > 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(deltas) + (size_t)iter * kDeltaRow +
                      (tile & (kDeltaCopies - 1)) * kNumSums + 31,
                  (unsigned long long)total);
    if (ps.active && tid == 64)
        atomicOr(ps.active + (size_t)(iter % kFlagRows) * ps.flag_words + (tile >> 6), 1ull << (tile & 63));
    // the first of the few points this tile had to search: where the next launch looks ahead
    if (APPLY && CERTIFY && fixed_tile < 0 && few && total > 0 && tid == 65) hints[tile] = queue[0];
    SVICP_STAMP(7);
}

// LARGE (sources of many thousands of tiles, in the launches that search most of their points;
// see icp_iterate_many_kernel for the others): the blocks run in many rounds, and a solve of
// 1.6 us in every block of every round costs more than one more launch, so a solve_kernel
// launched before this one has left R and T at rt_given.  The blocks take their tiles in the
// fixed order, and the flag rows (ps_active) are as long as the source needs.
template <bool APPLY, bool CERTIFY, int CH, bool INPLACE, bool LARGE = false>
__global__ __launch_bounds__(kItThreads, SVICP_TILE_OCC) void icp_iterate_kernel(
    // what the first loads of a launch need comes first
    fx_t *__restrict__ totals, fx_t *__restrict__ deltas, const int *__restrict__ fx, int iter, int n,
    unsigned long long *ps_active, Float3 *ps_tp, Float3 *ps_nv, float *ps_lb,   // a PointState, member by member
    const float *sx_in, const float *sy_in, const float *sz_in, float *sx_out, float *sy_out,
    float *sz_out, IcpState *st, double *X_hist, double *info_hist, CohParams cp,
    double *__restrict__ partials, const GridDesc *__restrict__ grid, const double *__restrict__ rt_given,
    float4 *ps_box)
{
    // The target's grid is described in device memory, not in the kernel's arguments: only the
    // search reads it (scalar loads where it needs them), and thirty words fewer have to be
    // fetched, and parked in spare register lanes, before a launch's first loads can go out.
    const GridDesc &g = *grid;
    const int nb = (int)gridDim.x;
    PointState ps;
    ps.tp = ps_tp; ps.nv = ps_nv; ps.lb = ps_lb; ps.active = ps_active;
    if (LARGE) { ps.flag_words = (nb + 63) >> 6; ps.box = ps_box; }
    const bool adaptive = !LARGE && APPLY && CERTIFY && ps.active != nullptr;
    iterate_tile<APPLY, CERTIFY, CH, INPLACE, LARGE>(g, sx_in, sy_in, sz_in, sx_out, sy_out, sz_out, n,
                                                     totals, deltas, fx, st, X_hist, info_hist, iter, cp, ps,
                                                     partials,
                                                     adaptive ? -1 : xcd_contiguous((int)blockIdx.x, nb), nb,
                                                     /* solve_now = */ !LARGE, rt_given);
}

// The same question for a tile, from the records its four waves left.
__device__ __forceinline__ bool tile_cannot_move(const float4 *rec, const double *rt)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    bool all_said = true;
#pragma unroll
    for (int w = 0; w < kItThreads / 64; ++w) {
        const float4 a = rec[w * kBoxWords], b = rec[w * kBoxWords + 1];
        // (a wave without live points leaves (+inf, -inf) too: at the end of the source only,
        // where the tile is then read)
        all_said = all_said && a.x <= b.x;
        lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
        hi[0] = fmaxf(hi[0], b.x); hi[1] = fmaxf(hi[1], b.y); hi[2] = fmaxf(hi[2], b.z);
    }
    double R[9], T[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rt[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) T[k] = rt[9 + k];
    return all_said && box_cannot_move(lo, hi, R, T);
}

// Sources of many thousands of tiles.  A block per tile runs such a source in many rounds of
// resident blocks, and late in a registration, when most tiles have nothing to do, every round
// still costs the few microseconds a block needs to find that out (16 rounds: 47 us for the
// 16.8 M-point pair, unchanged when 96 % of the tiles were not even read).  Here the launch is
// one round of blocks, each of which solves once and then works through several tiles:
//  * the tiles that had work in the launch before (a flag bit each, as for choose_tile) are
//    dealt out in turn -- the j-th of them to block j mod blocks -- because the work that is left
//    late in a registration sits in runs and at power-of-two strides of the sorted source, and
//    a block with several such tiles does them one after the other (8 to 10 us each);
//  * every other tile belongs to a fixed block (block b, in the order that keeps neighbours on one
//    XCD, has the `share` consecutive tiles from b * share on), which as a rule proves from the
//    tile's box record that it cannot have moved.  While the first wave solves, the second reads
//    the flags and the others fetch the box records of the block's fixed share, so that none of
//    this costs a trip to memory per tile.
// Whatever the flags say, every tile is worked on exactly once: by the block it is dealt to if
// its bit is set, by its fixed block if not (the row is not written during the launch).
constexpr int kManyRounds = 32;
constexpr int kManyFlagWords = 1024;    // flag words kept in LDS: 65536 tiles (beyond: fixed shares only)
__device__ __forceinline__ int many_tile(int first, int round, int share)
{
    return first * share + round;
}

template <bool APPLY, bool CERTIFY, int CH, bool INPLACE>
__global__ __launch_bounds__(kItThreads, SVICP_MANY_OCC) void icp_iterate_many_kernel(
    fx_t *__restrict__ totals, fx_t *__restrict__ deltas, const int *__restrict__ fx, int iter, int n, int nb,
    unsigned long long *ps_active, Float3 *ps_tp, Float3 *ps_nv, float *ps_lb, float4 *ps_box,
    const float *sx_in, const float *sy_in, const float *sz_in, float *sx_out, float *sy_out,
    float *sz_out, IcpState *st, double *X_hist, double *info_hist, CohParams cp,
    double *__restrict__ partials, const GridDesc *__restrict__ grid)
{
    __shared__ double rt[12];
    __shared__ float4 records[kManyRounds][kBoxRecord];
    __shared__ unsigned long long flags[kManyFlagWords];
    __shared__ int dealt[kManyRounds];
    __shared__ int n_dealt_shared;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int words = (nb + 63) >> 6;
    const int blocks = (int)gridDim.x;
    const int first = xcd_contiguous((int)blockIdx.x, blocks);
    const int share = (nb + blocks - 1) / blocks;      // (at most kManyRounds: the launch sees to that)
    const int rounds = max(0, min(min(share, nb - first * share), kManyRounds));
    const bool provable = APPLY && CERTIFY && INPLACE && ps_box != nullptr;
    const bool dealing = APPLY && ps_active != nullptr && words <= kManyFlagWords;
    if (APPLY) {
        if (wv == 0) {
            const SumsInFlight sums = request_sums(totals + (size_t)(iter - 1) * kNumSums,
                                                   deltas + (size_t)(iter - 1) * kDeltaRow, fx);
            solve_and_update(sums, totals + (size_t)iter * kNumSums, st, X_hist, info_hist, iter - 1,
                             blockIdx.x + 1 == gridDim.x, rt);
        } else if (wv == 1) {
            if (dealing) {
                // lane L looks after `per` consecutive words of the row the launch before wrote
                const unsigned long long *row = ps_active + (size_t)((iter - 1) % kFlagRows) * words;
                const int per = (words + 63) >> 6;
                int cnt = 0;
                for (int q = 0; q < per; ++q) {
                    const int w = lane * per + q;
                    const unsigned long long f = w < words ? row[w] : 0ull;
                    if (w < words) flags[w] = f;
                    cnt += __popcll(f);
                }
                const int inc = wave_inclusive_scan(cnt), excl = inc - cnt;
                if (lane == 0) n_dealt_shared = 0;
                // Of the tiles with a bit, counted through the row, this block has number me,
                // me + blocks, ...  While there are more of them than blocks, me follows the order
                // that keeps neighbouring tiles on one XCD (and its L2 on one stretch of the
                // target); once they fit into one round it is the block's own number, which the
                // dispatcher spreads over XCDs and compute units in turn -- in the other order
                // 679 tiles with work filled five XCDs four blocks deep and left three idle.
                const int with_work = __builtin_amdgcn_readlane(inc, 63);
                const int me = with_work <= blocks ? (int)blockIdx.x : first;
                for (int j = excl + ((me - excl) % blocks + blocks) % blocks; j < inc; j += blocks) {
                    int m = j - excl, tile = -1;
                    for (int q = 0; q < per && tile < 0; ++q) {
                        const unsigned long long f = flags[lane * per + q];
                        const int c = __popcll(f);
                        if (m < c) tile = 64 * (lane * per + q) + select64(f, m);
                        m -= c;
                    }
                    const int slot = atomicAdd(&n_dealt_shared, 1);
                    if (slot < kManyRounds) dealt[slot] = tile;
                }
            }
        } else if (provable) {
            for (int w = (int)threadIdx.x - 128; w < rounds * kBoxRecord; w += kItThreads - 128) {
                const int r = w / kBoxRecord;
                records[r][w % kBoxRecord] = ps_box[(size_t)many_tile(first, r, share) * kBoxRecord + w % kBoxRecord];
            }
        }
    }
    __syncthreads();
    // ---- the fixed share: lane r of every wave looks at tile r of it (every wave for itself, so
    // that the answer needs no further barrier).  A tile is left alone if it was dealt to some
    // block, or if its records prove that it stays where it is: what the tile routine would find
    // out from the coordinates (its step D) is known here without them -- once most of a large
    // cloud has stopped moving, reading coordinates is all such a launch would still do. ----
    bool mine_to_do = false, proven = false;
    if (lane < rounds) {
        const int t = many_tile(first, lane, share);
        const bool is_dealt = dealing && ((flags[t >> 6] >> (t & 63)) & 1ull);
        proven = !is_dealt && provable && tile_cannot_move(records[lane], rt);
        mine_to_do = !is_dealt && !proven;
    }
    const unsigned todo = (unsigned)__ballot(mine_to_do), n_proven = (unsigned)__popcll(__ballot(proven));
    // (what steers the loop is the same in every lane: told to the compiler, so that it lives in
    // scalar registers -- the tile routine needs every vector register it can get)
    const int n_dealt = dealing ? __builtin_amdgcn_readfirstlane(min(n_dealt_shared, kManyRounds)) : 0;
    const int n_todo = __popc(todo);
    PointState ps;
    ps.tp = ps_tp; ps.nv = ps_nv; ps.lb = ps_lb; ps.active = ps_active; ps.flag_words = words; ps.box = ps_box;
    // tile 0 may be among the proven ones: what its block does for the next launch is done here
    if (first == 0 && ((todo & 1u) == 0u) && !(dealing && (flags[0] & 1ull)))
        start_next_rows(deltas, ps_active, words, iter, 0);
    for (int i = 0; i < n_dealt + n_todo; ++i) {
        int tile;
        if (i < n_dealt) {
            tile = __builtin_amdgcn_readfirstlane(dealt[i]);
        } else {
            // the (i - n_dealt)-th set bit of todo
            unsigned m = todo;
            for (int k = n_dealt; k < i; ++k) m &= m - 1u;
            tile = many_tile(first, __ffs(m) - 1, share);
        }
        iterate_tile<APPLY, CERTIFY, CH, INPLACE, true>(*grid, sx_in, sy_in, sz_in, sx_out, sy_out, sz_out, n,
                                                        totals, deltas, fx, st, X_hist, info_hist, iter, cp, ps,
                                                        partials, tile, nb, /* solve_now = */ false, rt);
        __syncthreads();          // (the next tile uses the same shared arrays)
    }
    // slot 30 of the iteration's delta row counts the tiles left unread (a statistic for the
    // caller, like the searched points in slot 31)
    if (APPLY && threadIdx.x == 0 && n_proven > 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(deltas) + (size_t)iter * kDeltaRow +
                      (blockIdx.x & (kDeltaCopies - 1)) * kNumSums + 30,
                  (unsigned long long)n_proven);
}

// ------------------------------------------------------------------------
// Many independent registrations in one launch per iteration: block b works on
// tile block_map[b].y of pair block_map[b].x.  Each pair has its own grid,
// source, state and sums; nothing crosses pairs.
// ------------------------------------------------------------------------
struct BatchPair {
    GridDesc g;
    const float *x0, *y0, *z0;   // the source as set
    float *wx, *wy, *wz;         // the moving copy
    int n, nb;
    fx_t *totals, *deltas;
    const int *fx;
    IcpState *st;
    double *X_hist, *info_hist;
    CohParams cp;
    PointState ps;
    double *partials;
};

template <bool APPLY, bool CERTIFY, int CH>
__global__ __launch_bounds__(kItThreads, 4) void icp_iterate_batch_kernel(
    const BatchPair *__restrict__ pairs, const int2 *__restrict__ block_map, int iter,
    const double *__restrict__ rt_all)
{
    // rt_all: R and T of every pair, 12 doubles each, indexed by the pair's number so that
    // they can be asked for together with the pair's descriptor
    const int2 m = block_map[xcd_contiguous((int)blockIdx.x, (int)gridDim.x)];
    const BatchPair *P = pairs + m.x;
    const GridDesc g = P->g;
    const CohParams cp = P->cp;
    const PointState ps = P->ps;
    float *wx = P->wx, *wy = P->wy, *wz = P->wz;
    // iteration 0 reads the source as set and leaves it alone (the reference
    // applies the first update at the start of iteration 1), iteration 1 reads
    // it and writes the moving copy, later ones update the moving copy in place
    const bool first = iter <= 1;
    iterate_tile<APPLY, CERTIFY, CH>(g, first ? P->x0 : wx, first ? P->y0 : wy, first ? P->z0 : wz,
                                     wx, wy, wz, P->n, P->totals, P->deltas, P->fx, P->st, P->X_hist,
                                     P->info_hist, iter, cp, ps, P->partials, m.y, P->nb,
                                     /* solve_now = */ false, rt_all + 12 * (size_t)m.x);
}

// One block per pair: identity state, zero sums before iteration 0 and zero
// delta rows for `iters` iterations.
__global__ __launch_bounds__(64) void batch_reset_kernel(const BatchPair *__restrict__ pairs, int iters)
{
    const BatchPair *P = pairs + blockIdx.x;
    IcpState *st = P->st;
    const int t = threadIdx.x;
    if (t < 9) st->R[t] = (t % 4 == 0) ? 1.0 : 0.0;
    if (t < 3) st->T[t] = 0.0;
    if (t < 16) st->Trans[t] = (t % 5 == 0) ? 1.0 : 0.0;
    if (t < kNumSums) { st->S[t] = 0.0; P->totals[t] = 0; }
    if (t == 0) st->status = 0;
    for (int k = t; k < iters * kDeltaRow; k += 64) P->deltas[k] = 0;
}

// One wave per pair: the solve of iteration `iter` (before the launch of iteration iter + 1,
// and after the last one).  With a tile per block and 64 tiles per pair the blocks of an
// iteration would otherwise solve every pair's system 64 times over, several rounds of
// blocks deep, and a settled iteration would be little else.
__global__ __launch_bounds__(64) void batch_solve_kernel(const BatchPair *__restrict__ pairs, int iter,
                                                         double *__restrict__ rt_all)
{
    __shared__ double rt[12];
    const BatchPair *P = pairs + blockIdx.x;
    solve_and_update(request_sums(P->totals + (size_t)iter * kNumSums,
                                  P->deltas + (size_t)iter * kDeltaRow, P->fx),
                     P->totals + (size_t)(iter + 1) * kNumSums, P->st, P->X_hist, P->info_hist, iter,
                     true, rt);
    __syncthreads();
    if (threadIdx.x < 12) rt_all[12 * (size_t)blockIdx.x + threadIdx.x] = rt[threadIdx.x];
    for (int k = threadIdx.x; k < kDeltaRow; k += 64) P->deltas[k] = 0;   // clean for the pair's next run
}

// Gathers every pair's state into one array for a single copy to the host.
__global__ void batch_collect_kernel(const BatchPair *__restrict__ pairs, int count,
                                     IcpState *__restrict__ out)
{
    constexpr int W = sizeof(IcpState) / sizeof(double);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)count * W) return;
    const int p = (int)(i / W), w = (int)(i % W);
    reinterpret_cast<double *>(out + p)[w] = reinterpret_cast<const double *>(pairs[p].st)[w];
}

// What a run whose small buffers keep an earlier run's contents takes to be zero when it
// starts: row 0 of the totals and rows 0 and 1 of the per-iteration changes.  A kernel rather
// than two memsets, so that a recording of the run holds kernel nodes only.
__global__ __launch_bounds__(64) void clear_run_rows_kernel(fx_t *totals, fx_t *deltas)
{
    const int t = threadIdx.x;
    if (t < kNumSums) totals[t] = 0;
    for (int k = t; k < 2 * kDeltaRow; k += 64) deltas[k] = 0;
}

// One wave: solve the system of iteration `iter` and record the result (the
// last iteration of a run has no following iteration kernel to do it).
// dev_out (device memory, or null): the accumulated transform (16 doubles, row-major) and
// {status, inliers, sum w b^2, |X|} of the last iteration, kRunOutDoubles doubles in all.
constexpr int kRunOutDoubles = 20;
__global__ __launch_bounds__(64) void solve_kernel(fx_t *totals, fx_t *deltas, const int *fx,
                                                   IcpState *st, double *X_hist,
                                                   double *info_hist, int iter, IcpState *host_copy,
                                                   double *dev_out, double *rt_out = nullptr)
{
    __shared__ double rt[12];
    solve_and_update(request_sums(totals + (size_t)iter * kNumSums,
                                  deltas + (size_t)iter * kDeltaRow, fx),
                     totals + (size_t)(iter + 1) * kNumSums, st, X_hist, info_hist, iter, true,
                     rt);
    __syncthreads();
    if (rt_out) {
        // between two iterations of a run: R and T for the next launch, nothing else
        if (threadIdx.x < 12) rt_out[threadIdx.x] = rt[threadIdx.x];
        return;
    }
    // leave the first delta row clean for the next run, and hand the state to the
    // host through mapped memory (one store per lane instead of a copy command)
    for (int k = threadIdx.x; k < kDeltaRow; k += 64) deltas[k] = 0;
    if (host_copy || dev_out) __threadfence();
    if (host_copy) {
        constexpr int W = sizeof(IcpState) / sizeof(double);
        for (int k = threadIdx.x; k < W; k += 64)
            reinterpret_cast<double *>(host_copy)[k] = reinterpret_cast<volatile double *>(st)[k];
        __threadfence_system();
    }
    if (dev_out) {
        const volatile IcpState *v = st;
        const int t = threadIdx.x;
        if (t < 16) dev_out[t] = v->Trans[t];
        else if (t == 16) dev_out[16] = (double)v->status;
        else if (t == 17) dev_out[17] = v->S[27];
        else if (t == 18) dev_out[18] = v->S[28];
        else if (t == 19) dev_out[19] = v->S[30];
    }
}

// Sharded runs: this rank's change of the sums in one iteration, copies added up.
__global__ void fold_delta_kernel(const fx_t *__restrict__ delta_row, fx_t *__restrict__ out32)
{
    const int t = threadIdx.x;
    if (t >= kNumSums) return;
    fx_t s = 0;
    for (int cp = 0; cp < kDeltaCopies; ++cp) s += delta_row[cp * kNumSums + t];
    out32[t] = s;
}

// Sharded runs: the change summed over all ranks replaces the rank's own (copy 0 holds
// it, the other copies are cleared).
__global__ void unfold_delta_kernel(const fx_t *__restrict__ in32, fx_t *__restrict__ delta_row)
{
    for (int k = threadIdx.x; k < kDeltaRow; k += blockDim.x)
        delta_row[k] = k < kNumSums ? in32[k] : 0;
}

__global__ void reset_state_kernel(IcpState *st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < 9; ++k) st->R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; ++k) st->T[k] = 0.0;
        for (int k = 0; k < 16; ++k) st->Trans[k] = (k % 5 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < kNumSums; ++k) st->S[k] = 0.0;
        st->status = 0;
    }
}


// ------------------------------------------------------------------------
// Exact 1-NN far from the target: one candidate box per WAVE, staged in LDS.
//
// A query whose nearest target point lies beyond the probe radius has to look several cells
// out, and thread by thread that is 50 to 200 candidates per query gathered lane by lane, behind
// two table look-ups per row of cells.  But the 64 queries of a wave are neighbours (Morton
// order), and so are the boxes around them: the wave takes ONE box around all its open queries,
// copies the box's points into LDS once, and every lane runs over all staged points by broadcast
// reads, with the same separately rounded distance and the same (distance, index) order as
// everywhere else.
//
// The box is looked up block by block (4 x 4 x 4 cells): one directory word per block of the
// box, all in one trip; only the blocks that hold points go on (a surface fills a fraction of a
// box's blocks), each with its 16 rows of cells: a row's cells inside the box are one contiguous
// stretch of the sorted target (a "piece": start and end from the block's cell table, one
// more trip for all pieces of 4 blocks).  The pieces' points are dealt to the lanes 64 at a
// time, so the copies are coalesced and all in flight together.
//
// A lane is settled when its best distance lies inside the sphere around ITS query that the
// box is certain to contain (the test of ring_box, from the lane's own position); the others go
// on with a box that reaches as far as the largest best distance among them (then every lane
// settles), or, while some lane has found nothing at all, with a pad twice as wide.  Blocks
// that lie inside the box of the round before are left out (their points have been seen).  What
// the box cannot settle (too many blocks or points) is left to the caller's search by rings.
// ------------------------------------------------------------------------
constexpr int kFarCap = 256;            // points staged per wave and pass (4 KB)
constexpr int kFarMinOpen = 8;          // open lanes from which a wave goes straight to the list
constexpr int kFarMaxBlocks = 128;      // blocks of cells a box may have (more: the wave's queries are split up)
constexpr int kFarMaxRanked = 32;       // blocks looked up at a time (a lane each), all of which may hold points
constexpr int kFarMaxPieces = 16 * kFarMaxRanked;
constexpr int kFarMaxPoints = 2560;     // points a box may hold (more: the wave's queries are split up)
constexpr int kFarRounds = 6;
#ifdef SVICP_FAR_STATS
// development build: waves, rounds, blocks looked up, points staged, lanes entering, lanes settled,
// give-ups, -, then the time (10 ns units) of the four phases; summed in registers and added to
// the totals once, when a wave of nn1_far_kernel has finished all its work
__device__ unsigned long long g_far_stats[16];
struct FarCounters { unsigned long long v[12]; };
#define SVICP_FAR_COUNT(k, val) do { fc.v[k] += (unsigned long long)(val); } while (0)
#define SVICP_FAR_T(var) const unsigned long long var = wall_clock64()
#else
struct FarCounters { };
#define SVICP_FAR_COUNT(k, val) do { } while (0)
#define SVICP_FAR_T(var) do { } while (0)
#endif

__device__ __forceinline__ int wave_inclusive_max(int v);    // (defined with the k-NN kernels below)

// A wave's working memory in LDS (8.8 KB).
struct FarStage {
    float4 xy[kFarCap / 2];              // pairs of points {x0, x1, y0, y1}: the operands of the packed arithmetic
    float4 zw[kFarCap / 2];              //                 {z0, z1, index0, index1}
    unsigned short heads[kFarCap];       // per place of a pass: 1 + the piece that starts there, or 0
    unsigned piece_start[kFarMaxPieces]; // where a piece starts in the sorted target
    unsigned short piece_len[kFarMaxPieces], piece_excl[kFarMaxPieces];   // its points, and the points before it
    int2 ranked[kFarMaxRanked];          // {rank, block coordinates packed} of the blocks that hold points
};

// `open`: this lane has a finite query without a final answer.  On return `open` lanes that were
// settled have done = true and their answer in best / bidx (bidx = -1: the target is empty).
// Returns false when a box got too large (the open queries lie far apart: a wave that straddles
// a jump of the Morton order); the caller then offers the lanes still open part by part.
__device__ __forceinline__ bool far_search_wave(const GridDesc &g, float qx, float qy, float qz, bool open,
                                                float &best, int &bidx, bool &done, FarStage &st, FarCounters &fc,
                                                const int max_blocks = kFarMaxBlocks)
{
    const int lane = threadIdx.x & 63;
    constexpr unsigned long long kNoKey = ((unsigned long long)0x7f800000u << 32) | 0xffffffffu;
    unsigned long long bkey = kNoKey;
    if (open && best < INFINITY && bidx >= 0)
        bkey = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bidx;
    // round 0: one cell around the queries' cells; two where most of the wave is open (a stretch
    // of the source away from the target: one larger box then settles most lanes at once)
    float pad = __popcll(__ballot(open)) >= 32 ? 2.0f * g.h : g.h;
    const v2f q2x = { qx, qx }, q2y = { qy, qy }, q2z = { qz, qz };
    // the box of the round before (all of it has been scanned): empty before the first
    int ox0 = 0, ox1 = -1, oy0 = 0, oy1 = -1, oz0 = 0, oz1 = -1;
    float *const sxy = reinterpret_cast<float *>(st.xy), *const szw = reinterpret_cast<float *>(st.zw);
    SVICP_FAR_COUNT(0, 1);
    for (int round = 0; round < kFarRounds; ++round) {
        if (__ballot(open) == 0ull) return true;
        SVICP_FAR_COUNT(1, 1);
        // the box: cells that hold [lowest query - pad, highest query + pad] on every axis
        const float lx = wave_min_f(open ? qx : INFINITY), hx = wave_max_f(open ? qx : -INFINITY);
        const float ly = wave_min_f(open ? qy : INFINITY), hy = wave_max_f(open ? qy : -INFINITY);
        const float lz = wave_min_f(open ? qz : INFINITY), hz = wave_max_f(open ? qz : -INFINITY);
        auto cell = [&](float v, float o, int n) {
            float f = (v - o) * g.inv_h;
            f = fminf(fmaxf(f, 0.f), (float)(n - 1));
            return (int)f;
        };
        // (never smaller than the box before)
        int x0 = cell(lx - pad, g.ox, g.nx), x1 = cell(hx + pad, g.ox, g.nx);
        int y0 = cell(ly - pad, g.oy, g.ny), y1 = cell(hy + pad, g.oy, g.ny);
        int z0 = cell(lz - pad, g.oz, g.nz), z1 = cell(hz + pad, g.oz, g.nz);
        if (ox1 >= ox0) {
            x0 = min(x0, ox0); x1 = max(x1, ox1); y0 = min(y0, oy0); y1 = max(y1, oy1);
            z0 = min(z0, oz0); z1 = max(z1, oz1);
        }
        const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
        const int nbxs = (x1 >> 2) - bx0 + 1, nbys = (y1 >> 2) - by0 + 1, nbzs = (z1 >> 2) - bz0 + 1;
        const int nblk = nbxs * nbys * nbzs;
        if (nblk > max_blocks) { SVICP_FAR_COUNT(6, 1); return false; }
        SVICP_FAR_COUNT(2, nblk);
        // the radius around this lane's query the box is certain to cover (as ring_box)
        float rho = INFINITY;
        if (x0 > 0) rho = fminf(rho, qx - (g.ox + (float)x0 * g.h));
        if (x1 < g.nx - 1) rho = fminf(rho, (g.ox + (float)(x1 + 1) * g.h) - qx);
        if (y0 > 0) rho = fminf(rho, qy - (g.oy + (float)y0 * g.h));
        if (y1 < g.ny - 1) rho = fminf(rho, (g.oy + (float)(y1 + 1) * g.h) - qy);
        if (z0 > 0) rho = fminf(rho, qz - (g.oz + (float)z0 * g.h));
        if (z1 < g.nz - 1) rho = fminf(rho, (g.oz + (float)(z1 + 1) * g.h) - qz);
        const bool whole_grid = rho == INFINITY;
        if (!whole_grid) rho = rho * (1.0f - 1e-5f) - g.cov_slack;

        // ---- the blocks of the box that hold points and do not lie inside the old box ----
        // quotients by multiplication: (t + 0.5) / d is at least 0.5 / d from an integer and
        // t < 2^9, d <= 2^9, so the reciprocal's rounding cannot carry it across one
        const float inv_nbxs = __builtin_amdgcn_rcpf((float)nbxs), inv_nbys = __builtin_amdgcn_rcpf((float)nbys);
        bool gave_up = false;
        int seen = 0;
        for (int bb = 0; bb < nblk && !gave_up; bb += kFarMaxRanked) {
            SVICP_FAR_T(t0b);
            wave_lds_sync();                              // (the batch before has read its tables)
            const int t = lane < kFarMaxRanked ? bb + lane : nblk;
            const int row = (int)(((float)t + 0.5f) * inv_nbxs);
            const int bx = bx0 + t - row * nbxs;
            const int zr = (int)(((float)row + 0.5f) * inv_nbys);
            const int by = by0 + row - zr * nbys, bz = bz0 + zr;
            int rank = -1;
            if (t < nblk) {
                // (a block all of whose cells in the new box lie in the old box was scanned then)
                const bool inside_old = ox1 >= ox0 && max(bx * 4, x0) >= ox0 && min(bx * 4 + 3, x1) <= ox1 &&
                                        max(by * 4, y0) >= oy0 && min(by * 4 + 3, y1) <= oy1 &&
                                        max(bz * 4, z0) >= oz0 && min(bz * 4 + 3, z1) <= oz1;
                if (!inside_old) rank = g.dir[(bz * g.nby + by) * g.nbx + bx];
            }
            const unsigned long long has = __ballot(rank >= 0);
            const int nranked = __popcll(has);
            if (rank >= 0)
                st.ranked[__popcll(has & ((1ull << lane) - 1ull))] =
                    make_int2(rank, (bx - bx0) | ((by - by0) << 10) | ((bz - bz0) << 20));   // (each below 512)
            wave_lds_sync();
            SVICP_FAR_COUNT(8, wall_clock64() - t0b);
            if (nranked == 0) continue;
            SVICP_FAR_T(t1);
            // ---- pieces: the 16 rows of cells of every such block, clipped to the box; the loads of
            // four groups of 64 pieces are in flight together ----
            const int npieces = 16 * nranked;
            int total = 0;
            for (int pb = 0; pb < npieces; pb += 256) {
                unsigned s4[4], e4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = pb + 64 * u + lane;
                    s4[u] = 0; e4[u] = 0;
                    if (pc < npieces) {
                        const int2 blk = st.ranked[pc >> 4];
                        const int cellrow = pc & 15;
                        const int pbx = bx0 + (blk.y & 1023), pby = by0 + ((blk.y >> 10) & 1023), pbz = bz0 + (blk.y >> 20);
                        const int y = pby * 4 + (cellrow & 3), z = pbz * 4 + (cellrow >> 2);
                        if (y >= y0 && y <= y1 && z >= z0 && z <= z1) {
                            int xa = max(x0, pbx * 4), xb = min(x1, pbx * 4 + 3);
                            // the cells of this row the old box held have been scanned: cut them off
                            // where they are at one end of the piece (a piece that reaches beyond
                            // the old box on both sides is taken whole)
                            if (ox1 >= ox0 && y >= oy0 && y <= oy1 && z >= oz0 && z <= oz1) {
                                if (xa >= ox0) xa = max(xa, ox1 + 1);
                                else if (xb <= ox1) xb = min(xb, ox0 - 1);
                            }
                            if (xa <= xb) {
                                const int base = blk.x * 64 + (cellrow << 2);
                                s4[u] = g.run2[base + (xa & 3)].x;
                                e4[u] = g.end1[base + (xb & 3)];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = pb + 64 * u + lane;
                    if (pb + 64 * u >= npieces) break;
                    const int len = (int)(e4[u] - s4[u]);
                    const int incl = wave_inclusive_scan(len);
                    if (pc < npieces) {
                        st.piece_start[pc] = s4[u];
                        st.piece_len[pc] = (unsigned short)min(len, 65535);
                        st.piece_excl[pc] = (unsigned short)min(total + incl - len, 65535);
                    }
                    total += __builtin_amdgcn_readlane(incl, 63);
                }
            }
            seen += total;
            if (total > 65535 || seen > kFarMaxPoints) { gave_up = true; break; }
            SVICP_FAR_COUNT(3, total);
            SVICP_FAR_COUNT(9, wall_clock64() - t1);

            // ---- passes of kFarCap points: dealt to the lanes, staged, scanned ----
            for (int base = 0; base < total; base += kFarCap) {
                const int here = min(total - base, kFarCap);
                SVICP_FAR_T(t2);
                wave_lds_sync();            // the pass before has been read
#pragma unroll
                for (int c = 0; c < kFarCap / 64; ++c)
                    if (c * 64 < here) st.heads[c * 64 + lane] = 0;
                wave_lds_sync();
                // every piece with points in this pass marks its first place there
                for (int pc = lane; pc < npieces; pc += 64) {
                    const int len = st.piece_len[pc], excl = st.piece_excl[pc];
                    if (len > 0 && excl + len > base && excl < base + kFarCap)
                        st.heads[max(excl - base, 0)] = (unsigned short)(pc + 1);
                }
                wave_lds_sync();
                float4 pt[kFarCap / 64];
                int piece_before = 0;
#pragma unroll
                for (int c = 0; c < kFarCap / 64; ++c) {
                    if (c * 64 >= here) break;
                    const int p = c * 64 + lane;
                    const int h = max(wave_inclusive_max((int)st.heads[p]), piece_before);
                    piece_before = __builtin_amdgcn_readlane(h, 63);
                    const int pc = max(h - 1, 0);
                    // (a place past the end reads its piece's first point again: dropped below)
                    const unsigned src = st.piece_start[pc] + (unsigned)(p < here ? base + p - (int)st.piece_excl[pc] : 0);
                    pt[c] = g.pts[src];
                }
#pragma unroll
                for (int c = 0; c < kFarCap / 64; ++c) {
                    if (c * 64 >= here) break;
                    const int slot = c * 64 + lane, pr = slot >> 1, el = slot & 1;
                    if (slot < here) {
                        sxy[4 * pr + el] = pt[c].x; sxy[4 * pr + 2 + el] = pt[c].y;
                        szw[4 * pr + el] = pt[c].z; szw[4 * pr + 2 + el] = pt[c].w;
                    }
                }
                // fill the last group of four with points that win nothing
                if (lane < 4) {
                    const int slot = here + lane;
                    if (slot < ((here + 3) & ~3)) {
                        const int pr = slot >> 1, el = slot & 1;
                        sxy[4 * pr + el] = INFINITY; sxy[4 * pr + 2 + el] = INFINITY;
                        szw[4 * pr + el] = INFINITY; szw[4 * pr + 2 + el] = __uint_as_float(0xffffffffu);
                    }
                }
                wave_lds_sync();
                SVICP_FAR_COUNT(10, wall_clock64() - t2);
                SVICP_FAR_T(t3);
                for (int pr = 0; pr < (here + 1) / 2; pr += 2) {
                    const float4 a0 = st.xy[pr], b0 = st.zw[pr], a1 = st.xy[pr + 1], b1 = st.zw[pr + 1];
                    v2f dx = q2x - (v2f){ a0.x, a0.y }, dy = q2y - (v2f){ a0.z, a0.w }, dz = q2z - (v2f){ b0.x, b0.y };
                    v2f d0 = dx * dx; d0 = d0 + dy * dy; d0 = d0 + dz * dz;
                    dx = q2x - (v2f){ a1.x, a1.y }; dy = q2y - (v2f){ a1.z, a1.w }; dz = q2z - (v2f){ b1.x, b1.y };
                    v2f d1 = dx * dx; d1 = d1 + dy * dy; d1 = d1 + dz * dz;
                    unsigned long long k;
                    k = ((unsigned long long)__float_as_uint(d0.x) << 32) | __float_as_uint(b0.z); bkey = k < bkey ? k : bkey;
                    k = ((unsigned long long)__float_as_uint(d0.y) << 32) | __float_as_uint(b0.w); bkey = k < bkey ? k : bkey;
                    k = ((unsigned long long)__float_as_uint(d1.x) << 32) | __float_as_uint(b1.z); bkey = k < bkey ? k : bkey;
                    k = ((unsigned long long)__float_as_uint(d1.y) << 32) | __float_as_uint(b1.w); bkey = k < bkey ? k : bkey;
                }
                SVICP_FAR_COUNT(11, wall_clock64() - t3);
            }
        }
        if (gave_up) { SVICP_FAR_COUNT(6, 1); return false; }
        // who is settled: the best distance lies inside the sphere the box covers, or the box
        // is the whole grid (then an empty answer is final too)
        const bool found = (unsigned)(bkey >> 32) < 0x7f800000u;
        const float bd = __uint_as_float((unsigned)(bkey >> 32));
        if (open && (whole_grid || (found && rho > 0.f && bd <= rho * rho))) {
            done = true; open = false;
            best = found ? bd : INFINITY;
            bidx = found ? (int)(unsigned)bkey : -1;
        }
        if (whole_grid) return true;
        ox0 = x0; ox1 = x1; oy0 = y0; oy1 = y1; oz0 = z0; oz1 = z1;
        // next box: as far as the farthest best distance among the open lanes (every one of them
        // then settles); while some lane has seen no point at all, twice the pad
        const bool blind = __ballot(open && !found) != 0ull;
        const float reach = wave_max_f(open && found ? sqrtf(bd) : 0.f);
        pad = blind ? 2.0f * pad : fmaxf(reach * (1.0f + 1e-4f) + 2.0f * g.cov_slack + 1e-3f * g.h, pad);
        // (a box that reaches `reach` beyond every open query leaves rho >= reach for each)
    }
    return __ballot(open) == 0ull;      // (rounds used up with lanes still open: as a box too large)
}

// The exact search of one query by rings of cells around its probe origin, each lane for
// itself (the best so far carries over; later rings scan only the shell around the box already
// seen).  Returns false when no ring up to kMaxRing settles the query: the exhaustive pass then.
__device__ __forceinline__ bool ring_search_lane(const GridDesc &g, float qx, float qy, float qz, float &best,
                                                 int &bpos, int &bidx, const int max_ring = kMaxRing)
{
    int cx, cy, cz;
    clamped_origin(g, qx, qy, qz, cx, cy, cz);
    auto visit = [&](unsigned j, const float4 &p) {
        float d = dist2(qx, qy, qz, p);
        int id = __float_as_int(p.w);
        if (better(d, id, best, bidx)) { best = d; bpos = (int)j; bidx = id; }
    };
    // the box of the previous ring (already scanned): empty before the first scan
    int px0 = 0, px1 = -1, py0 = 0, py1 = -1, pz0 = 0, pz1 = -1;
    for (int ring = 0; ring <= max_ring; ++ring) {
        int x0, x1, y0, y1, z0, z1;
        float rho = ring_box(g, qx, qy, qz, cx, cy, cz, ring, x0, x1, y0, y1, z0, z1);
        if (px1 < px0) {
            scan_box_points(g, x0, x1, y0, y1, z0, z1, visit);
        } else {
            if (z0 < pz0) scan_box_points(g, x0, x1, y0, y1, z0, pz0 - 1, visit);
            if (z1 > pz1) scan_box_points(g, x0, x1, y0, y1, pz1 + 1, z1, visit);
            if (y0 < py0) scan_box_points(g, x0, x1, y0, py0 - 1, pz0, pz1, visit);
            if (y1 > py1) scan_box_points(g, x0, x1, py1 + 1, y1, pz0, pz1, visit);
            if (x0 < px0) scan_box_points(g, x0, px0 - 1, py0, py1, pz0, pz1, visit);
            if (x1 > px1) scan_box_points(g, px1 + 1, x1, py0, py1, pz0, pz1, visit);
        }
        px0 = x0; px1 = x1; py0 = y0; py1 = y1; pz0 = z0; pz1 = z1;
        if (rho == INFINITY || (bpos >= 0 && rho > 0.f && best <= rho * rho)) return true;
    }
    return false;
}

// ------------------------------------------------------------------------
// Standalone 1-NN over the resident source (CKNNSearch::getKNeighbour with
// k = 1, CKNNSearch.cpp:67-71).  EXACT = false: neighbours within the probe
// radius only, as the iteration loop searches.  EXACT = true: unbounded;
// queries the ring search cannot settle are queued for the brute-force pass.
// ------------------------------------------------------------------------
template <bool EXACT, int CH>
__global__ __launch_bounds__(kBlockThreads) void nn1_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, const int *__restrict__ perm, int n, float thr,
    int *__restrict__ idx_out, float *__restrict__ d2_out, int *__restrict__ unresolved,
    int *__restrict__ n_unresolved, int4 *__restrict__ far_list = nullptr, int *__restrict__ n_far = nullptr)
{
    // Capped search: a block takes consecutive chunks of 256 queries, and the blocks of one
    // XCD (blockIdx % 8 equal) consecutive blocks' worth, so each XCD's L2 sees one stretch of
    // the source, as in the iteration kernel.  Exact search: chunks dealt out one by one; its
    // time goes to the queries far from the target, which come in stretches, and they are
    // better spread over all blocks.
    const long long chunks = ((long long)n + kBlockThreads - 1) / kBlockThreads;
    const long long per_block = (chunks + gridDim.x - 1) / gridDim.x;
    const long long first = EXACT ? (long long)blockIdx.x
                                  : (long long)xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * per_block;
    const long long last = EXACT ? chunks : (first + per_block < chunks ? first + per_block : chunks);
    for (long long c = first; c < last; c += EXACT ? (long long)gridDim.x : 1) {
        const long long i_raw = c * kBlockThreads + threadIdx.x;
        // (the exact search keeps whole waves together: a lane past the end repeats the last
        // query and writes nothing)
        const bool inrange = i_raw < n;
        if (!EXACT && !inrange) break;
        const long long i = inrange ? i_raw : (long long)n - 1;
        float qx = qx_[i], qy = qy_[i], qz = qz_[i];
        float best; int bpos, bidx;
        nn_capped<CH>(g, qx, qy, qz, best, bpos, bidx);
        const long long o = perm ? perm[i] : i;
        if (!EXACT) {
            bool hit = bpos >= 0 && best <= thr;
            if ((kDrop & kDropOut) && best != 12345.678f) continue;    // (the search stays alive)
            idx_out[o] = hit ? bidx : -1;
            d2_out[o] = hit ? best : INFINITY;
            continue;
        }
        // The capped probe has seen every target point in the cube [q - rr, q + rr] (its runs
        // span the quarter cells that hold the cube, not the whole 2 x 2 x 2 cells): a best
        // distance inside rr is final.  Anything else starts over with the box of ring 0 in
        // full (the probe's candidates are met again; the best so far carries over).
        float r0 = g.rr * (1.0f - 1e-5f) - g.cov_slack;
        {
            // what the probe's runs really span: five quarter cells in x from the quarter of
            // q - rr, two cells in y and in z; the sphere around q inside that box
            int ax, ay, az, aq;
            if (probe_origin(g, qx, qy, qz, ax, ay, az, &aq)) {
                const float hq = g.h * (1.0f / (float)kFineX);
                float rho = fminf(qx - (g.ox + (float)aq * hq), (g.ox + (float)(aq + kFineRun) * hq) - qx);
                rho = fminf(rho, fminf(qy - (g.oy + (float)ay * g.h), (g.oy + (float)(ay + 2) * g.h) - qy));
                rho = fminf(rho, fminf(qz - (g.oz + (float)az * g.h), (g.oz + (float)(az + 2) * g.h) - qz));
                r0 = fmaxf(r0, rho * (1.0f - 1e-5f) - g.cov_slack);
            }
        }
        bool done = bpos >= 0 && r0 > 0.f && best <= r0 * r0;
        // Lanes without an answer.  A few of them in a wave (the usual case once the clouds are
        // registered: the nearest point lies just beyond the probe radius): each scans the whole
        // 2 x 2 x 2 cells around its probe origin for itself (ring 0).  Many (a stretch of the source
        // away from the target), or whoever ring 0 does not settle: the wave goes on the list of nn1_far_kernel,
        // which searches one box of candidates for all its open lanes.
        const bool finite = fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY;
        bool open = inrange && !done && finite && g.m > 0;
        const bool listed = far_list != nullptr;
        const bool few_open = __popcll(__ballot(open)) < kFarMinOpen;
        if (open && (!listed || few_open)) {
            done = ring_search_lane(g, qx, qy, qz, best, bpos, bidx, listed ? 0 : kMaxRing);
            open = !done && listed;
        }
        // (the list has an entry per wave of this kernel, at the wave's own place: nothing to count)
        const unsigned long long open_mask = __ballot(open);
        if (listed && (threadIdx.x & 63) == 0)
            far_list[i_raw >> 6] = make_int4((int)(i_raw), (int)(unsigned)open_mask, (int)(unsigned)(open_mask >> 32), 0);
        if (!inrange || open) continue;
        // (not finite, or an empty target: nothing is nearest)
        if (!done && listed) { done = true; bpos = -1; }
        if (done) {
            idx_out[o] = bpos >= 0 ? bidx : -1;
            d2_out[o] = bpos >= 0 ? best : INFINITY;
        } else {
            int slot = atomicAdd(n_unresolved, 1);
            unresolved[slot] = (int)i;
        }
    }
}

// The waves nn1_kernel<true> put on its list: one wave here per listed wave there, the same 64
// queries, of which those marked open share a candidate box (far_search_wave).  What the box does
// not settle searches by rings, and what they do not settle goes to the exhaustive pass.
constexpr int kFarThreads = 128;        // two waves, 17.5 KB of LDS
// Where a box gets too large (the open queries lie far apart: a wave that straddles a jump of the
// Morton order) the wave's open lanes go on the next list in `parts` parts of `part_lanes` Morton
// neighbours each (16 x 4, then 4 x 1), for the next launch of this kernel, so that the parts are spread
// over the GPU instead of being worked off one after the other by the wave that failed.  Without
// a next list (the last launch: single queries) they go to the exhaustive pass.
__global__ __launch_bounds__(kFarThreads, 4) void nn1_far_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, const int *__restrict__ perm, int n, const int4 *__restrict__ far_list,
    const int *__restrict__ n_far, int far_capacity, int4 *__restrict__ next_list, int *__restrict__ n_next, int next_capacity,
    int parts, int part_lanes, int max_blocks, int *__restrict__ idx_out, float *__restrict__ d2_out,
    int *__restrict__ unresolved, int *__restrict__ n_unresolved)
{
    constexpr int W = kFarThreads / 64;
    __shared__ FarStage stage[W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (the first list has an entry per wave of the search kernel: no counter; a list that ran
    // over was not written beyond its capacity)
    const int count = n_far ? min(*n_far, far_capacity) : far_capacity;
    FarCounters fc = {};
    for (int w = (int)blockIdx.x * W + wv; w < count; w += (int)gridDim.x * W) {
        const int4 entry = far_list[w];
        const unsigned long long mask = (unsigned long long)(unsigned)entry.y | ((unsigned long long)(unsigned)entry.z << 32);
        const long long i_raw = (long long)entry.x + lane;
        const bool open = (mask >> lane) & 1ull;            // (open lanes are in range)
        const long long i = i_raw < n ? i_raw : (long long)n - 1;
        const float qx = qx_[i], qy = qy_[i], qz = qz_[i];
        float best = INFINITY;
        int bidx = -1;
        bool done = false;
        const bool ok = far_search_wave(g, qx, qy, qz, open, best, bidx, done, stage[wv], fc, max_blocks);
        bool passed_on = false;
        if (!ok && next_list) {
            // the lanes still open, in four parts, to the next launch
            const unsigned long long left = __ballot(open && !done);
            // (`parts` parts of `part_lanes` lanes: the lanes this entry's mask can span)
            int slot = 0;
            if (lane == 0) slot = atomicAdd(n_next, parts);
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (slot + parts <= next_capacity) {
                passed_on = true;
                if (lane < parts) {
                    const int low = (int)(__ffsll((long long)mask) - 1) & ~(parts * part_lanes - 1);
                    const unsigned long long span = part_lanes >= 64 ? ~0ull : (1ull << part_lanes) - 1ull;
                    const unsigned long long m = left & (span << (low + lane * part_lanes));
                    next_list[slot + lane] = make_int4(entry.x, (int)(unsigned)m, (int)(unsigned)(m >> 32), 0);
                }
            }
        }
        if (!open || (passed_on && !done)) continue;
        int bpos = bidx >= 0 ? 0 : -1;
        SVICP_FAR_COUNT(7, __popcll(__ballot(!done)));
        // (what no box settled, down to a box of its own, lies far from every target point: the
        // exhaustive pass takes it)
        const long long o = perm ? perm[i] : i;
        if (done) {
            idx_out[o] = bpos >= 0 ? bidx : -1;
            d2_out[o] = bpos >= 0 ? best : INFINITY;
        } else {
            int slot = atomicAdd(n_unresolved, 1);
            unresolved[slot] = (int)i;
        }
    }
#ifdef SVICP_FAR_STATS
    // (SVICP_FAR_STATS = 2: the launches after the first only)
    if (lane == 0 && (SVICP_FAR_STATS != 2 || n_far != nullptr))
        for (int k = 0; k < 12; ++k) atomicAdd(&g_far_stats[k], fc.v[k]);
#endif
}

// ------------------------------------------------------------------------
// Gather probe (measurement only): the vector-memory instruction stream of the capped
// search with nothing else around it, to see what the texture-address path of this GPU
// delivers for it.  REAL: every query makes the same table look-ups (directory, its
// probe record) and loads exactly the candidates nn_capped would, four at a
// time as the search does, no distance arithmetic, no running best, nothing written per query.  !REAL:
// `per_query` 16-byte loads per lane from hashed places of the sorted target (no look-ups,
// no locality between lanes), eight in flight.  With `instr` non-null every wave adds the
// vector-memory instructions it issued to *instr.
// ------------------------------------------------------------------------
template <bool REAL, int BYTES = 16>
__global__ __launch_bounds__(kBlockThreads) void gather_probe_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, int n, int per_query, unsigned *__restrict__ sink,
    unsigned long long *__restrict__ instr)
{
    const long long chunks = ((long long)n + kBlockThreads - 1) / kBlockThreads;
    const long long per_block = (chunks + gridDim.x - 1) / gridDim.x;
    const long long first = (long long)xcd_contiguous((int)blockIdx.x, (int)gridDim.x) * per_block;
    const long long last = first + per_block < chunks ? first + per_block : chunks;
    unsigned acc = 0;
    unsigned long long issued = 0;      // wave-level count (uniform)
    for (long long c = first; c < last; ++c) {
        const long long i = c * kBlockThreads + threadIdx.x;
        // REAL with per_query = s > 1: only every s-th lane takes part (what a gather costs
        // per active lane rather than per instruction)
        const bool live = i < n && (!REAL || per_query <= 1 || (threadIdx.x % per_query) == 0);
        const long long ic = live ? i : 0;
        const float qx = qx_[ic], qy = qy_[ic], qz = qz_[ic];
        issued += 3;
        unsigned T = 0, P1 = 0, P2 = 0, P3 = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        if (REAL) {
            int cx, cy, cz, cxq;
            uint2 run[4] = { make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u) };
            if (live && probe_origin(g, qx, qy, qz, cx, cy, cz, &cxq)) {
                const int rank = g.dir[dir_index(g, cx, cy, cz)];
                const size_t e = rank >= 0 ? (size_t)rank * (64 * kFineX) + quarter_local(cxq, cy, cz) : 0;
                uint2 got[4];
                probe_runs(g, e, got);
                if (rank >= 0) {
                    run[0] = got[0]; run[1] = got[1]; run[2] = got[2]; run[3] = got[3];
                }
            }
            issued += 2;
            P1 = run[0].y - run[0].x;
            P2 = P1 + (run[1].y - run[1].x);
            P3 = P2 + (run[2].y - run[2].x);
            T = P3 + (run[3].y - run[3].x);
            o0 = run[0].x; o1 = run[1].x - P1; o2 = run[2].x - P2; o3 = run[3].x - P3;
        } else {
            T = live ? (unsigned)per_query : 0u;
        }
        // groups of G loads (the search: 4); a lane whose list has ended drops out, as in the search
        constexpr int G = REAL ? 4 : 8;
        if (instr) {      // the counting launch only: the wave walks as many groups as its longest lane
            unsigned Tmax = T;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) Tmax = max(Tmax, (unsigned)__shfl_xor((int)Tmax, off, 64));
            issued += (unsigned long long)((Tmax + G - 1) / G) * G;
        }
        for (unsigned f = 0; f < T; f += G) {
            float4 p[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const unsigned ff = f + k;
                unsigned j;
                if (REAL) {
                    j = ff + (ff < P1 ? o0 : ff < P2 ? o1 : ff < P3 ? o2 : o3);
                    j = ff < T ? j : 0u;
                } else {
                    unsigned hsh = ((unsigned)i * 2654435761u) ^ (ff * 0x9E3779B9u);
                    hsh ^= hsh >> 15; hsh *= 0x85EBCA6Bu; hsh ^= hsh >> 13;
                    j = ff < T ? hsh % (unsigned)g.m : 0u;
                }
                // BYTES < 16: records of that size from the packed array of normals (sorted like
                // the points, so the same neighbourhoods), to see what a narrower gather costs
                if (BYTES == 16) p[k] = g.pts[j];
                else if (BYTES == 12) { const Float3 t = reinterpret_cast<const Float3 *>(g.nrm)[j]; p[k] = make_float4(t.x, t.y, t.z, 0.f); }
                else if (BYTES == 8) { const float2 t = reinterpret_cast<const float2 *>(g.nrm)[j]; p[k] = make_float4(t.x, t.y, 0.f, 0.f); }
                else p[k] = make_float4(g.nrm[j], 0.f, 0.f, 0.f);
            }
#pragma unroll
            // all four words are used, so that the loads stay 16-byte loads
            for (int k = 0; k < G; ++k)
                acc ^= (__float_as_uint(p[k].w) + __float_as_uint(p[k].x)) ^ (__float_as_uint(p[k].y) + __float_as_uint(p[k].z));
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;      // keeps the loads alive
    // (same-address atomics serialise at about 6 ns each: counted in a launch of its own)
    if (instr && (threadIdx.x & 63) == 0) atomicAdd(instr, issued);
}

// One block per queued query: scan the whole target.
__global__ __launch_bounds__(kBlockThreads) void nn1_brute_kernel(
    GridDesc g, const float *__restrict__ qx_, const float *__restrict__ qy_,
    const float *__restrict__ qz_, const int *__restrict__ perm,
    const int *__restrict__ unresolved, const int *__restrict__ n_unresolved,
    int *__restrict__ idx_out, float *__restrict__ d2_out)
{
    __shared__ float sd[kBlockThreads];
    __shared__ int si[kBlockThreads];
    const int cnt = *n_unresolved;
    for (int u = blockIdx.x; u < cnt; u += gridDim.x) {
        const int i = unresolved[u];
        float qx = qx_[i], qy = qy_[i], qz = qz_[i];
        float best = INFINITY; int bidx = 0x7fffffff;
        for (int j = threadIdx.x; j < g.m; j += kBlockThreads) {
            float4 p = g.pts[j];
            float d = dist2(qx, qy, qz, p);
            int id = __float_as_int(p.w);
            if (better(d, id, best, bidx)) { best = d; bidx = id; }
        }
        sd[threadIdx.x] = best; si[threadIdx.x] = bidx;
        __syncthreads();
        for (int s = kBlockThreads / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s &&
                better(sd[threadIdx.x + s], si[threadIdx.x + s], sd[threadIdx.x], si[threadIdx.x])) {
                sd[threadIdx.x] = sd[threadIdx.x + s];
                si[threadIdx.x] = si[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const long long o = perm ? perm[i] : i;
            // (a query with a NaN coordinate is nearer to nothing: no neighbour)
            const bool any = g.m > 0 && si[0] != 0x7fffffff;
            idx_out[o] = any ? si[0] : -1;
            d2_out[o] = any ? sd[0] : INFINITY;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------
// Exact k-NN, 1 <= k <= KMAX (CKNNSearch::getKNeighbour, CKNNSearch.cpp:67-71).
// One thread per query keeps the k best (distance, index) pairs sorted in
// private memory (indexed by a variable, so it lives in scratch: the slow, general path).
// ------------------------------------------------------------------------
template <int KMAX>
struct KList {
    float d[KMAX];
    int id[KMAX];
    int cnt, k;
    __device__ void clear(int k_) { cnt = 0; k = k_; }
    __device__ void insert(float dd, int ii)
    {
        if (cnt == k && !better(dd, ii, d[k - 1], id[k - 1])) return;
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && better(dd, ii, d[pos - 1], id[pos - 1])) {
            d[pos] = d[pos - 1]; id[pos] = id[pos - 1]; --pos;
        }
        d[pos] = dd; id[pos] = ii;
        if (cnt < k) ++cnt;
    }
};

template <int KMAX, bool LIST>
__global__ __launch_bounds__(kBlockThreads) void knn_kernel(
    GridDesc g, const float *__restrict__ q_aos, int n, int k, int *__restrict__ idx_out,
    float *__restrict__ d2_out, const int *__restrict__ list, const int *__restrict__ n_list,
    int *__restrict__ unresolved, int *__restrict__ n_unresolved)
{
    // LIST: the queries named in `list` (those the wave search below passed on)
    const int total = LIST ? *n_list : n;
    const int first = k > 1 ? 1 : 0;
    for (long long t = (long long)blockIdx.x * kBlockThreads + threadIdx.x; t < total;
         t += (long long)gridDim.x * kBlockThreads) {
        const long long i = LIST ? list[t] : t;
        float qx = q_aos[3 * i], qy = q_aos[3 * i + 1], qz = q_aos[3 * i + 2];
        KList<KMAX> L;
        L.clear(k);
        bool done = false;
        int cx, cy, cz;
        clamped_origin(g, qx, qy, qz, cx, cy, cz);
        for (int ring = first; ring <= kKnnMaxRing; ++ring) {
            int x0, x1, y0, y1, z0, z1;
            float rho = ring_box(g, qx, qy, qz, cx, cy, cz, ring, x0, x1, y0, y1, z0, z1);
            L.clear(k);
            scan_box(g, x0, x1, y0, y1, z0, z1, [&](unsigned j) {
                float4 p = g.pts[j];
                L.insert(dist2(qx, qy, qz, p), __float_as_int(p.w));
            });
            if (rho == INFINITY || (L.cnt == k && rho > 0.f && L.d[k - 1] <= rho * rho)) {
                done = true;
                break;
            }
        }
        if (done) {
            for (int j = 0; j < k; ++j) {
                idx_out[i * k + j] = j < L.cnt ? L.id[j] : -1;
                d2_out[i * k + j] = j < L.cnt ? L.d[j] : INFINITY;
            }
        } else {
            int slot = atomicAdd(n_unresolved, 1);
            unresolved[slot] = (int)i;
        }
    }
}

// ------------------------------------------------------------------------
// Exact k-NN, k <= kKnnWaveK, by one whole wave per query (the same answers as
// knn_kernel: the k smallest (distance, index) keys of a ring's cell box, accepted
// when the k-th lies within the radius the box is certain to cover, else the next ring).
//
// A box is rows of cells along x, each row at most one contiguous run of the sorted
// target per block of cells it crosses.  64 runs at a time, one per lane: a prefix sum
// over the run lengths numbers the candidates, and the candidates are dealt to the
// lanes 64 at a time (run heads marked in LDS, spread by a prefix maximum), so the
// loads are contiguous and no lane idles on a short run.  Candidates are taken in
// rounds of kKnnWaveChunks x 64; after each round the wave keeps the at most 64
// smallest keys, one per lane: a bisection on the distance bits finds a bound t with
// k <= #(distance <= t) <= 64 (counting is a ballot and a population count), the keys
// within t are packed through LDS, and at the end every kept key counts the kept keys
// below it, which is its place in the answer.  More than 64 - k + 1 candidates at one
// distance defeat the bisection: such a query goes on `ties` for knn_kernel<., true>;
// one that the last ring does not settle goes on `far` for the exhaustive pass.  The
// lists are appended to 64 entries at a time (one atomic per 64: same-address atomics
// serialise, and every query of a cloud far from the target ends on a list).
// ------------------------------------------------------------------------
constexpr int kKnnWaveK = 32;
constexpr int kKnnWaveChunks = 8;
constexpr int kKnnWaveRound = 64 * kKnnWaveChunks;

__device__ __forceinline__ int wave_inclusive_max(int v)   // v >= 0 in every lane
{
    v = max(v, dpp_get<0x111, 0xf, true>(v));
    v = max(v, dpp_get<0x112, 0xf, true>(v));
    v = max(v, dpp_get<0x114, 0xf, true>(v));
    v = max(v, dpp_get<0x118, 0xf, true>(v));
    v = max(v, dpp_get<0x142, 0xa, false>(v));
    v = max(v, dpp_get<0x143, 0xc, false>(v));
    return v;
}

// A wave's share of a list that many waves append to: lane j holds the j-th pending entry.
struct WaveAppender {
    int pending, count;
    __device__ void flush(int *list, int *n_list)
    {
        if (count == 0) return;
        int slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(n_list, count);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if ((int)(threadIdx.x & 63) < count) list[slot + (threadIdx.x & 63)] = pending;
        count = 0;
    }
    __device__ void add(int v, int *list, int *n_list)   // v the same in every lane
    {
        if ((int)(threadIdx.x & 63) == count) pending = v;
        if (++count == 64) flush(list, n_list);
    }
};

__global__ __launch_bounds__(kBlockThreads) void knn_wave_kernel(
    GridDesc g, const float *__restrict__ q_aos, int n, int k, int *__restrict__ idx_out,
    float *__restrict__ d2_out, int *__restrict__ ties, int *__restrict__ n_ties,
    int *__restrict__ far, int *__restrict__ n_far)
{
    constexpr int W = kBlockThreads / 64;
    constexpr unsigned long long kNone = ~0ull;          // sorts after every real key
    constexpr unsigned kNanBits = 0x7fc00000u;           // every real distance's bits are <= this
    __shared__ int head[W][kKnnWaveRound];
    __shared__ unsigned long long picked[W][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *const my_head = head[wv];
    unsigned long long *const my_picked = picked[wv];
    WaveAppender to_ties = { 0, 0 }, to_far = { 0, 0 };

    for (long long i = (long long)blockIdx.x * W + wv; i < n; i += (long long)gridDim.x * W) {
        const float qx = q_aos[3 * i], qy = q_aos[3 * i + 1], qz = q_aos[3 * i + 2];
        int cx, cy, cz;
        clamped_origin(g, qx, qy, qz, cx, cy, cz);
        bool done = false, gave_up = false;
        unsigned long long best = kNone;     // the kept keys, one per lane, in no order
        int kept = 0, place = 0;

        for (int ring = (k > 1 ? 1 : 0); ring <= kKnnMaxRing && !done && !gave_up; ++ring) {
            int x0, x1, y0, y1, z0, z1;
            const float rho = ring_box(g, qx, qy, qz, cx, cy, cz, ring, x0, x1, y0, y1, z0, z1);
            const int bx0 = x0 >> 2, nbxs = (x1 >> 2) - bx0 + 1, ny = y1 - y0 + 1;
            const int nseg = nbxs * ny * (z1 - z0 + 1);
            // quotients by multiplication: (s + 0.5) / d is at least 0.5 / d from an integer and
            // s < 2^11, d <= 18, so the reciprocal's rounding cannot carry it across one
            const float inv_nbxs = __builtin_amdgcn_rcpf((float)nbxs), inv_ny = __builtin_amdgcn_rcpf((float)ny);
            best = kNone;
            kept = 0;
            unsigned last_t = kNanBits;          // every kept key's distance bits are <= this
            for (int sg = 0; sg < nseg && !gave_up; sg += 64) {
                // this lane's run: segment (z, y, block in x)
                unsigned s = 0;
                int len = 0;
                if (sg + lane < nseg) {
                    const int row = (int)(((float)(sg + lane) + 0.5f) * inv_nbxs);
                    const int bx = bx0 + (sg + lane) - row * nbxs;
                    const int zr = (int)(((float)row + 0.5f) * inv_ny), y = y0 + row - zr * ny, z = z0 + zr;
                    const int rank = g.dir[((z >> 2) * g.nby + (y >> 2)) * g.nbx + bx];
                    if (rank >= 0) {
                        const int xa = max(x0, bx * 4), xb = min(x1, bx * 4 + 3);
                        const int base = rank * 64 + (((z & 3) << 4) | ((y & 3) << 2));
                        s = g.run2[base + (xa & 3)].x;
                        len = (int)(g.end1[base + (xb & 3)] - s);
                    }
                }
                const int incl = wave_inclusive_scan(len), excl = incl - len;
                const int T = __builtin_amdgcn_readlane(incl, 63);

                for (int base = 0; base < T; base += kKnnWaveRound) {
                    const int here = min(T - base, kKnnWaveRound);
                    // run heads of this round; the run under way at `base` heads position 0
                    wave_lds_sync();
#pragma unroll
                    for (int c = 0; c < kKnnWaveChunks; ++c) {
                        if (c * 64 >= here) break;
                        my_head[c * 64 + lane] = 0;
                    }
                    wave_lds_sync();
                    if (len > 0 && incl > base && excl < base + kKnnWaveRound)
                        my_head[max(excl - base, 0)] = lane + 1;
                    wave_lds_sync();

                    unsigned long long key[kKnnWaveChunks];
#pragma unroll
                    for (int c = 0; c < kKnnWaveChunks; ++c) key[c] = kNone;
                    int run_before = 0;
#pragma unroll
                    for (int c = 0; c < kKnnWaveChunks; ++c) {
                        if (c * 64 >= here) break;
                        {
                            const int p = c * 64 + lane;
                            const int h = max(wave_inclusive_max(my_head[p]), run_before);
                            run_before = __builtin_amdgcn_readlane(h, 63);
                            const int seg = max(h - 1, 0);
                            const unsigned s_seg = (unsigned)__shfl((int)s, seg, 64);
                            const int excl_seg = __shfl(excl, seg, 64);
                            if (p < here) {
                                const float4 pt = g.pts[s_seg + (unsigned)(base + p - excl_seg)];
                                const float d = dist2(qx, qy, qz, pt);
                                const unsigned bits = d == d ? __float_as_uint(d) : kNanBits;
                                key[c] = ((unsigned long long)bits << 32) | __float_as_uint(pt.w);
                            }
                        }
                    }

                    // bound t: k <= #(distance bits <= t) <= 64 over this round and the kept keys
                    auto count_within = [&](unsigned t) {
                        int cnt = __popcll(__ballot((unsigned)(best >> 32) <= t));
#pragma unroll
                        for (int c = 0; c < kKnnWaveChunks; ++c) {
                            if (c * 64 >= here) break;
                            cnt += __popcll(__ballot((unsigned)(key[c] >> 32) <= t));
                        }
                        return cnt;
                    };
                    unsigned t = kNanBits;
                    if (here + kept > 64) {
                        // the first tries are guesses: on a surface the count grows about linearly
                        // with the squared distance, so one count tells where the middle of the
                        // window lies; then plain bisection of the bit patterns
                        long long lo = -1, hi = kNanBits;      // #(<= lo) < k, #(<= hi) > 64
                        float guess = kept > 0 && last_t < 0x7f800000u ? __uint_as_float(last_t)
                                      : rho < INFINITY                  ? rho * rho
                                                                        : 0.f;
                        const float aim = 0.5f * (float)(k + 64);
                        for (int tries = 0;; ++tries) {
                            if (hi - lo < 2) { gave_up = true; break; }
                            long long mid = (lo + hi) >> 1;
                            if (tries < 4 && guess > 0.f) {
                                const long long gb = __float_as_uint(guess);
                                if (gb > lo && gb < hi) mid = gb;
                            }
                            const int cnt = count_within((unsigned)mid);
                            if (cnt < k) lo = mid;
                            else if (cnt > 64) hi = mid;
                            else { t = (unsigned)mid; break; }
                            guess = __uint_as_float((unsigned)mid) *
                                    (cnt > 0 ? aim * __builtin_amdgcn_rcpf((float)cnt) : 4.f);
                        }
                        if (gave_up) break;
                    }
                    last_t = t;
                    // pack the keys within t: one per lane
                    int pos = 0;
                    auto pack = [&](unsigned long long v) {
                        const bool in = (unsigned)(v >> 32) <= t;
                        const unsigned long long m = __ballot(in);
                        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        if (in) my_picked[pos + before] = v;
                        pos += __popcll(m);
                    };
                    pack(best);
#pragma unroll
                    for (int c = 0; c < kKnnWaveChunks; ++c) {
                        if (c * 64 >= here) break;
                        pack(key[c]);
                    }
                    wave_lds_sync();
                    kept = pos;
                    best = lane < kept ? my_picked[lane] : kNone;
                }
            }
            if (gave_up) break;
            // fewer kept keys, fewer steps below: a second bound with k .. k + 4 keys within it,
            // when the distances allow one (one compare and one count per try)
            if (kept > k + 4) {
                long long lo = -1, hi = last_t;            // #(<= lo) < k, #(<= hi) > k + 4
                float guess = last_t < 0x7f800000u
                                  ? __uint_as_float(last_t) * ((float)(k + 2) * __builtin_amdgcn_rcpf((float)kept))
                                  : 0.f;
                for (int tries = 0; hi - lo >= 2; ++tries) {
                    long long mid = (lo + hi) >> 1;
                    if (tries < 2 && guess > 0.f) {
                        const long long gb = __float_as_uint(guess);
                        if (gb > lo && gb < hi) mid = gb;
                    }
                    const int cnt = __popcll(__ballot((unsigned)(best >> 32) <= (unsigned)mid));
                    if (cnt < k) lo = mid;
                    else if (cnt > k + 4) hi = mid;
                    else {
                        // pack again: the keys within the bound to lanes 0 .. cnt - 1
                        const bool in = (unsigned)(best >> 32) <= (unsigned)mid;
                        const unsigned long long m = __ballot(in);
                        wave_lds_sync();
                        if (in) my_picked[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = best;
                        wave_lds_sync();
                        kept = cnt;
                        best = lane < kept ? my_picked[lane] : kNone;
                        break;
                    }
                    guess = __uint_as_float((unsigned)mid) *
                            (cnt > 0 ? (float)(k + 2) * __builtin_amdgcn_rcpf((float)cnt) : 4.f);
                }
            }
            // place of each kept key among the kept keys (distinct: the index is part of a key)
            // (lanes from `kept` on hold kNone, which is below nothing: four at a time)
            place = 0;
            for (int j = 0; j < kept; j += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)best, j + u);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(best >> 32), j + u);
                    place += (((unsigned long long)hi << 32) | lo) < best ? 1 : 0;
                }
            }
            if (rho == INFINITY) {
                done = true;
            } else if (kept >= k && rho > 0.f) {
                const unsigned long long m = __ballot(best != kNone && place == k - 1);
                const int who = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                const float dk = __uint_as_float(__builtin_amdgcn_readlane((unsigned)(best >> 32), who));
                done = dk <= rho * rho;
            }
        }
        if (done) {
            const int have = min(kept, k);
            if (best != kNone && place < k) {
                idx_out[i * k + place] = (int)(unsigned)best;
                d2_out[i * k + place] = __uint_as_float((unsigned)(best >> 32));
            }
            if (lane < k - have) {
                idx_out[i * k + have + lane] = -1;
                d2_out[i * k + have + lane] = INFINITY;
            }
        } else if (gave_up) {
            to_ties.add((int)i, ties, n_ties);
        } else {
            to_far.add((int)i, far, n_far);
        }
    }
    to_ties.flush(ties, n_ties);
    to_far.flush(far, n_far);
}

// Exhaustive pass of the k-NN for the queries the ring search could not settle: one
// block per query.  Every thread keeps the k best of its share of the target; the k
// answers are then drawn one at a time from the heads of the 256 sorted lists (block-wide
// minimum of the (distance, index) keys, which are distinct).
template <int KMAX>
__global__ __launch_bounds__(kBlockThreads) void knn_brute_block_kernel(
    GridDesc g, const float *__restrict__ q_aos, int k, int *__restrict__ idx_out,
    float *__restrict__ d2_out, const int *__restrict__ unresolved,
    const int *__restrict__ n_unresolved)
{
    __shared__ unsigned long long wave_min[kBlockThreads / 64];
    const int cnt = *n_unresolved;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int u = blockIdx.x; u < cnt; u += gridDim.x) {
        const long long i = unresolved[u];
        const float qx = q_aos[3 * i], qy = q_aos[3 * i + 1], qz = q_aos[3 * i + 2];
        KList<KMAX> L;
        L.clear(k);
        for (int j = threadIdx.x; j < g.m; j += kBlockThreads) {
            const float4 p = g.pts[j];
            L.insert(dist2(qx, qy, qz, p), __float_as_int(p.w));
        }
        int head = 0;
        for (int r = 0; r < k; ++r) {
            const unsigned long long mine =
                head < L.cnt ? ((unsigned long long)__float_as_uint(L.d[head]) << 32) | (unsigned)L.id[head]
                             : ~0ull;
            unsigned long long m = mine;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(m, off, 64);
                m = o < m ? o : m;
            }
            if (lane == 0) wave_min[wv] = m;
            __syncthreads();
            unsigned long long best = wave_min[0];
#pragma unroll
            for (int w = 1; w < kBlockThreads / 64; ++w) best = wave_min[w] < best ? wave_min[w] : best;
            if (mine == best && best != ~0ull) ++head;
            if (threadIdx.x == 0) {
                idx_out[i * k + r] = best != ~0ull ? (int)(unsigned)best : -1;
                d2_out[i * k + r] = best != ~0ull ? __uint_as_float((unsigned)(best >> 32)) : INFINITY;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------
// Normal estimation from the k-NN lists.
// Restates CPoint2PlaneICP::calcNormals (CPoint2PlaneICP.cpp:130-152): raw
// second moments about the origin, each term fl32(fl32(pj*pi)/k) added in
// fp64 in neighbour order, eigenvector of the smallest eigenvalue (cyclic
// Jacobi on the 3x3), stored as fp32.  The sign of a normal is not defined by
// the reference's eigen-solver and does not affect the registration.
// centred (an option the reference does not have): the covariance of the neighbourhood about
// its centroid, in fp64, instead of the moments about the origin.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void normals_kernel(const float *__restrict__ ref_aos,
                                                                const int *__restrict__ knn_idx,
                                                                int m, int k, bool centred,
                                                                float *__restrict__ normals)
{
    const long long p = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
    if (p >= m) return;
    double W[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    double mean[3] = { 0, 0, 0 };
    const float kf = (float)k;
    if (centred) {
        for (int t = 0; t < k; ++t) {
            const size_t at = 3 * (size_t)knn_idx[p * k + t];
            mean[0] += (double)ref_aos[at]; mean[1] += (double)ref_aos[at + 1]; mean[2] += (double)ref_aos[at + 2];
        }
        mean[0] /= (double)k; mean[1] /= (double)k; mean[2] /= (double)k;
    }
    // four neighbours at a time: their indices, then their coordinates, are asked for together
    // (one after the other, each neighbour costs two round trips to memory); the sums still
    // take the terms in neighbour order
    for (int t0 = 0; t0 < k; t0 += 4) {
        int id[4];
        float c[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = t0 + u < k ? knn_idx[p * k + t0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t at = 3 * (size_t)max(id[u], 0);
            c[u][0] = ref_aos[at]; c[u][1] = ref_aos[at + 1]; c[u][2] = ref_aos[at + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (id[u] < 0) continue;
            if (centred) {
                const double d[3] = { (double)c[u][0] - mean[0], (double)c[u][1] - mean[1],
                                      (double)c[u][2] - mean[2] };
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) W[j][i] += d[j] * d[i];
                continue;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float term = __fdiv_rn(__fmul_rn(c[u][j], c[u][i]), kf);
                    W[j][i] += (double)term;
                }
        }
    }
    double V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = W[0][1] * W[0][1] + W[0][2] * W[0][2] + W[1][2] * W[1][2];
        double diag = W[0][0] * W[0][0] + W[1][1] * W[1][1] + W[2][2] * W[2][2];
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = a + 1; b < 3; ++b) {
                double apq = W[a][b];
                if (apq == 0.0) continue;
                double theta = (W[b][b] - W[a][a]) / (2.0 * apq);
                double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double wa = W[r][a], wb = W[r][b];
                    W[r][a] = cs * wa - sn * wb;
                    W[r][b] = sn * wa + cs * wb;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double wa = W[a][r], wb = W[b][r];
                    W[a][r] = cs * wa - sn * wb;
                    W[b][r] = sn * wa + cs * wb;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double va = V[r][a], vb = V[r][b];
                    V[r][a] = cs * va - sn * vb;
                    V[r][b] = sn * va + cs * vb;
                }
            }
    }
    double e0 = W[0][0], e1 = W[1][1], e2 = W[2][2];
    double v0, v1, v2;
    if (e0 <= e1 && e0 <= e2) { v0 = V[0][0]; v1 = V[1][0]; v2 = V[2][0]; }
    else if (e1 <= e2)        { v0 = V[0][1]; v1 = V[1][1]; v2 = V[2][1]; }
    else                      { v0 = V[0][2]; v1 = V[1][2]; v2 = V[2][2]; }
    double nrm = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
    normals[3 * p] = (float)(v0 / nrm);
    normals[3 * p + 1] = (float)(v1 / nrm);
    normals[3 * p + 2] = (float)(v2 / nrm);
}

// ------------------------------------------------------------------------
// grid construction
// ------------------------------------------------------------------------
__global__ void bbox_kernel(const float *__restrict__ aos, long long n, float *__restrict__ out6)
{
    // out6: per block {minx, miny, minz, maxx, maxy, maxz}
    __shared__ float s[6][kBlockThreads];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < n;
         i += (long long)gridDim.x * kBlockThreads)
        for (int a = 0; a < 3; ++a) {
            float v = aos[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; ++a) { s[a][threadIdx.x] = mn[a]; s[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int st = kBlockThreads / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int a = 0; a < 3; ++a) {
                s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + st]);
                s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out6[6 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
}

// The sort key of a target point: its cell (z-major, x fastest) times kFineX plus the quarter
// of the cell its x falls in, i.e. ((cz ny + cy) nx) kFineX + cxq with cxq the quarter's
// number along x.  key / kFineX is the cell's key.
__device__ __forceinline__ unsigned long long point_key(const GridDesc &g, float x, float y, float z)
{
    const int nxq = g.nx * kFineX;
    const int cxq = min(max((int)floorf((x - g.ox) * g.inv_h * (float)kFineX), 0), nxq - 1);
    const int cy = min(max((int)floorf((y - g.oy) * g.inv_h), 0), g.ny - 1);
    const int cz = min(max((int)floorf((z - g.oz) * g.inv_h), 0), g.nz - 1);
    return ((unsigned long long)cz * g.ny + cy) * nxq + cxq;
}

__device__ __forceinline__ void point_cell(const GridDesc &g, float x, float y, float z, int &cx,
                                           int &cy, int &cz)
{
    cx = min(max((int)floorf((x - g.ox) * g.inv_h), 0), g.nx - 1);
    cy = min(max((int)floorf((y - g.oy) * g.inv_h), 0), g.ny - 1);
    cz = min(max((int)floorf((z - g.oz) * g.inv_h), 0), g.nz - 1);
}

__global__ void cell_key_kernel(GridDesc g, const float *__restrict__ aos, int m,
                                unsigned long long *__restrict__ keys, int *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    keys[i] = point_key(g, aos[3 * (size_t)i], aos[3 * (size_t)i + 1], aos[3 * (size_t)i + 2]);
    vals[i] = i;
}

// marks the block of every occupied cell and of the cells one step below it in x, y
// and/or z: the cells from which a 2x2x2 probe reaches it
__global__ void mark_blocks_kernel(GridDesc g, const unsigned long long *__restrict__ keys, int m,
                                   int *__restrict__ marks)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long key = keys[i] / kFineX;          // the cell's key
    if (i > 0 && keys[i - 1] / kFineX == key) return;
    int cx = (int)(key % g.nx);
    unsigned long long rest = key / g.nx;
    int cy = (int)(rest % g.ny), cz = (int)(rest / g.ny);
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a)
                if (cx - a >= 0 && cy - b >= 0 && cz - c >= 0)
                    marks[dir_index(g, cx - a, cy - b, cz - c)] = 1;
}

__global__ void finish_dir_kernel(const int *__restrict__ marks, const int *__restrict__ ranks,
                                  int nblocks_total, int *__restrict__ dir,
                                  int *__restrict__ block_of_rank)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks_total) return;
    if (marks[b]) { dir[b] = ranks[b]; block_of_rank[ranks[b]] = b; }
    else dir[b] = -1;
}

__device__ __forceinline__ unsigned lower_bound_key(const unsigned long long *keys, int m,
                                                    unsigned long long key)
{
    int lo = 0, hi = m;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (unsigned)lo;
}

// First position at or after `lo` whose key is not below `key`, for a position expected
// close to `lo`: doubling steps, then bisection of the last step.  keys[lo - 1] < key.
__device__ __forceinline__ unsigned lower_bound_from(const unsigned long long *keys, unsigned lo,
                                                     int m, unsigned long long key)
{
    unsigned step = 16;
    while (lo + step < (unsigned)m && keys[lo + step] < key) { lo += step + 1; step *= 2; }
    unsigned hi = min(lo + step, (unsigned)m);
    while (lo < hi) {
        unsigned mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per cell of every ranked block: where the cell's points start and end in the
// sorted target, and where the next cell's in x end (one search of the whole array, two
// short ones from its result).
__global__ void fill_cells_kernel(GridDesc g, const unsigned long long *__restrict__ keys, int m,
                                  const int *__restrict__ block_of_rank, int nranked,
                                  uint2 *__restrict__ run2, unsigned *__restrict__ end1)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nranked * 64) return;
    int rank = (int)(t >> 6), local = (int)(t & 63);
    int b = block_of_rank[rank];
    int bx = b % g.nbx, by = (b / g.nbx) % g.nby, bz = b / (g.nbx * g.nby);
    int cx = bx * 4 + (local & 3), cy = by * 4 + ((local >> 2) & 3), cz = bz * 4 + (local >> 4);
    unsigned s = 0, e1 = 0, e2 = 0;
    if (cx < g.nx && cy < g.ny && cz < g.nz) {
        // (the keys carry the quarter of the cell in their low bits)
        unsigned long long key = (((unsigned long long)cz * g.ny + cy) * g.nx + cx) * kFineX;
        s = lower_bound_key(keys, m, key);
        e1 = lower_bound_from(keys, s, m, key + kFineX);
        e2 = (cx + 1 < g.nx) ? lower_bound_from(keys, e1, m, key + 2 * kFineX) : e1;
    }
    run2[t] = make_uint2(s, e2);
    end1[t] = e1;
}

// One thread per quarter cell of every ranked block: the run of kFineRun quarters that starts
// at it (found from the start of its cell, which fill_cells_kernel wrote).
__global__ void fill_quarter_runs_kernel(GridDesc g, const unsigned long long *__restrict__ keys, int m,
                                         const int *__restrict__ block_of_rank, int nranked,
                                         const uint2 *__restrict__ run2, uint2 *__restrict__ qrun)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nranked * 64 * kFineX) return;
    const int rank = (int)(t / (64 * kFineX)), local = (int)(t % (64 * kFineX));
    const int b = block_of_rank[rank];
    const int bx = b % g.nbx, by = (b / g.nbx) % g.nby, bz = b / (g.nbx * g.nby);
    const int cxq = bx * 4 * kFineX + (local & (4 * kFineX - 1)), cy = by * 4 + ((local >> kFineBits) & 3),
              cz = bz * 4 + (local >> (kFineBits + 2));
    const int nxq = g.nx * kFineX;
    unsigned s = 0, e = 0;
    if (cxq < nxq && cy < g.ny && cz < g.nz) {
        const unsigned long long row = ((unsigned long long)cz * g.ny + cy) * nxq;
        const unsigned cell_start = run2[(long long)rank * 64 + cell_local(cxq / kFineX, cy, cz)].x;
        s = lower_bound_from(keys, cell_start, m, row + cxq);
        e = lower_bound_from(keys, s, m, row + min(cxq + kFineRun, nxq));
    }
    qrun[t] = make_uint2(s, e);
}

// The four runs a probe from each quarter cell walks: its own run of kFineRun quarters and
// those of the same quarter in the rows next to it in y, in z and in both.  A row outside the
// grid or in an unranked block (no point in its run) gives an empty run.
__global__ void fill_quads_kernel(GridDesc g, const int *__restrict__ block_of_rank, int nranked,
                                  const uint2 *__restrict__ qrun, uint4 *__restrict__ quad,
                                  uint4 *__restrict__ quad_big, int big_capacity, int *__restrict__ big_count)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nranked * 64 * kFineX) return;
    const int rank = (int)(t / (64 * kFineX)), local = (int)(t % (64 * kFineX));
    const int b = block_of_rank[rank];
    const int bx = b % g.nbx, by = (b / g.nbx) % g.nby, bz = b / (g.nbx * g.nby);
    const int cxq = bx * 4 * kFineX + (local & (4 * kFineX - 1)), cy = by * 4 + ((local >> kFineBits) & 3),
              cz = bz * 4 + (local >> (kFineBits + 2));
    uint2 nb[4] = { qrun[t], make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u) };
    for (int r = 1; r < 4; ++r) {
        const int y = cy + (r & 1), z = cz + (r >> 1);
        if (cxq < g.nx * kFineX && y < g.ny && z < g.nz) {
            const int nrank = g.dir[dir_index(g, cxq / kFineX, y, z)];
            if (nrank >= 0) nb[r] = qrun[(long long)nrank * (64 * kFineX) + quarter_local(cxq, y, z)];
        }
    }
    // the packed form (probe_runs): an empty run takes its partner's start, so the difference
    // of the starts is that of two rows next to each other in y, or zero
    const unsigned l0 = nb[0].y - nb[0].x, l1 = nb[1].y - nb[1].x, l2 = nb[2].y - nb[2].x,
                   l3 = nb[3].y - nb[3].x;
    const unsigned s0 = l0 ? nb[0].x : nb[1].x, s1 = l1 ? nb[1].x : s0;
    const unsigned s2 = l2 ? nb[2].x : nb[3].x, s3 = l3 ? nb[3].x : s2;
    const bool fits = s1 >= s0 && s3 >= s2 && s1 - s0 <= 0xffffu && s3 - s2 <= 0xffffu && l0 <= 254u &&
                      l1 <= 254u && l2 <= 254u && l3 <= 254u;
    if (fits) {
        quad[t] = make_uint4(s0, s2, (s1 - s0) | ((s3 - s2) << 16), l0 | (l1 << 8) | (l2 << 16) | (l3 << 24));
    } else {
        // (which record gets which slot depends on the order of arrival; the slot is looked
        // at by nobody but probe_runs)
        const int slot = atomicAdd(big_count, 1);
        if (slot < big_capacity) {
            quad_big[2 * (size_t)slot] = make_uint4(nb[0].x, nb[0].y, nb[1].x, nb[1].y);
            quad_big[2 * (size_t)slot + 1] = make_uint4(nb[2].x, nb[2].y, nb[3].x, nb[3].y);
        }
        quad[t] = make_uint4((unsigned)slot, 0u, 0u, kQuadBig);
    }
}

__global__ void gather_target_kernel(const float *__restrict__ aos, const float *__restrict__ nrm_aos,
                                     const int *__restrict__ vals, int m, float4 *__restrict__ pts,
                                     float *__restrict__ nrm)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int o = vals[i];
    pts[i] = make_float4(aos[3 * (size_t)o], aos[3 * (size_t)o + 1], aos[3 * (size_t)o + 2],
                         __int_as_float(o));
    if (nrm_aos) {
        nrm[3 * (size_t)i] = nrm_aos[3 * (size_t)o];
        nrm[3 * (size_t)i + 1] = nrm_aos[3 * (size_t)o + 1];
        nrm[3 * (size_t)i + 2] = nrm_aos[3 * (size_t)o + 2];
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned v)
{
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// Morton key of each source point over the source's own bounding box: kMortonBits per axis
// (8192 steps: far finer than the spacing of any cloud dense enough to fill a tile, and the
// sort's cost goes with the number of key bits)
constexpr int kMortonBits = 13;
constexpr int kMortonMax = (1 << kMortonBits) - 1;
__global__ void morton_key_kernel(const float *__restrict__ aos, int n, float ox, float oy,
                                  float oz, float scale, unsigned long long *__restrict__ keys,
                                  int *__restrict__ vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [&](float v, float o) {
        float f = (v - o) * scale;
        f = fminf(fmaxf(f, 0.f), (float)kMortonMax);
        return (unsigned)f;
    };
    keys[i] = spread21(q(aos[3 * (size_t)i], ox)) | (spread21(q(aos[3 * (size_t)i + 1], oy)) << 1) |
              (spread21(q(aos[3 * (size_t)i + 2], oz)) << 2);
    vals[i] = i;
}

// Storage order of the source: Morton order, with the full 1024-point tiles dealt out so
// that the contiguous stretch of tiles an XCD works on (xcd_contiguous) is made of groups
// of 8 Morton-neighbouring tiles taken from all over the cloud, every eighth group (4, 16
// and 32 per group measured within 1 % of it).
// Where only part of the source lies near the target, the XCDs then share the searching
// instead of the one that owns that part doing most of it; inside a group the tiles stay
// neighbours, which is what the L2 needs.  sorted[p]: original index of the point at Morton
// position p; perm[q]: original index of the point stored at q.
constexpr int kDealGroup = 8;
__global__ void deal_tiles_kernel(const int *__restrict__ sorted, int n, int *__restrict__ perm)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int tile = p / kItTile, o = p % kItTile;
    const int whole = (n / kItTile) & ~(8 * kDealGroup - 1);   // full tiles, in rounds of 8 groups
    int q = p;
    if (tile < whole) {
        const int group = tile / kDealGroup, r = tile % kDealGroup;
        const int xcd = group & 7, k = (group >> 3) * kDealGroup + r;
        q = (xcd * (whole >> 3) + k) * kItTile + o;
    }
    perm[q] = sorted[p];
}

__global__ void iota_kernel(int *__restrict__ vals, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vals[i] = i;
}

__global__ void aos_to_soa_kernel(const float *__restrict__ aos, const int *__restrict__ perm, int n,
                                  float *__restrict__ x, float *__restrict__ y, float *__restrict__ z)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t o = perm ? (size_t)perm[i] : (size_t)i;
    x[i] = aos[3 * o]; y[i] = aos[3 * o + 1]; z[i] = aos[3 * o + 2];
}

__global__ void soa_to_aos_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                  const float *__restrict__ z, const int *__restrict__ perm, int n,
                                  float *__restrict__ aos)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t o = perm ? (size_t)perm[i] : (size_t)i;
    aos[3 * o] = x[i]; aos[3 * o + 1] = y[i]; aos[3 * o + 2] = z[i];
}

}  // namespace svicp
