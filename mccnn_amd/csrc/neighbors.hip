// Fixed-radius neighbour search over the 27-cell window and the per-edge kernel density
// estimate. Replaces tf_ops/find_neighbors.cu and tf_ops/compute_pdf.cu.
#include "batch.h"
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <type_traits>

namespace mccnn {

struct CentreCtx {
    float cx, cy, cz, R, T;  // T: squared-distance threshold equivalent to sqrt(d2) < R (common.h)
    int b, x, y, z;
};

__device__ __forceinline__ CentreCtx centre_ctx(const float* __restrict__ centres, const int* __restrict__ cb,
                                                const float* __restrict__ mn, const float* __restrict__ mx,
                                                int i, int B, int nc, float radius, int scaleInv, float Tabs) {
    CentreCtx c;
    c.b = clamp_batch(cb[i], B);
    c.cx = centres[(size_t)i * 3];
    c.cy = centres[(size_t)i * 3 + 1];
    c.cz = centres[(size_t)i * 3 + 2];
    float ext = max_extent(mn, mx, c.b);
    float cs = ext / (float)nc;
    c.R = scaleInv ? radius * ext : radius;  // find_neighbors.cu:73
    c.T = scaleInv ? sqrt_threshold(c.R) : Tabs;  // an absolute radius has ONE threshold: the host computes it
    c.x = cell_coord(c.cx, mn[c.b * 3], cs, nc);
    c.y = cell_coord(c.cy, mn[c.b * 3 + 1], cs, nc);
    c.z = cell_coord(c.cz, mn[c.b * 3 + 2], cs, nc);
    return c;
}

// One wave per 8 consecutive centres of the visiting order (`order`, identity when null; a cell-coherent order --
// the inverse sort permutation, or the centres' own counting-sort order in this grid -- makes neighbouring centres share a cell, and never
// changes results). Centres of one grid cell share their 27-cell window: the
// wave stages the window's candidates ONCE in LDS (canonical order: table order of find_neighbors.cu:282-291, ascending
// j inside a cell; the neighbour index travels in the .w lane of the staged point), then tests every centre of that
// cell against it with lanes = candidates: conflict-free LDS reads instead of one global gather per (centre,
// candidate) -- the per-centre walks issue 18 M float4 gathers per pass on the 100k room, the window form ~1 M -- and
// ballot / mbcnt compaction writes the hits in canonical order without per-cell counts. Earlier forms, measured on the
// 100k room (count + fill): one thread per centre 130 us, three threads per centre (one per z-slab) 37 + 60 us, a
// thread per (centre, cell) 60 + 82 us, this one 31 + 37 us. FILL == false counts per centre, FILL == true writes the
// (j, i) rows at startIdx[i].
// Windows larger than MCCNN_NW_CAP points are processed in segments of the flat candidate list.
// v[lane L] = val (wave-uniform), L a compile-time lane: v_writelane_b32 (one SGPR operand per instruction on gfx9, so the
// lane select is an inline constant)
template <int L>
__device__ __forceinline__ void writelane_u(unsigned& v, unsigned val) {
    const unsigned sv = __builtin_amdgcn_readfirstlane(val);
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sv), "n"(L));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
#ifndef MCCNN_NW_G
#define MCCNN_NW_G 8
#endif
#ifndef MCCNN_NW_CAP
#define MCCNN_NW_CAP 256
#endif
// Hit masks the count pass leaves for the fill pass: MCCNN_NW_ROUNDS 64-candidate rounds per centre (windows of up to
// 512 points; larger ones are searched again by the fill pass).
#ifndef MCCNN_NW_ROUNDS
#define MCCNN_NW_ROUNDS 8
#endif
#define MCCNN_NW_SCAN_M 4096   // lists of up to this many centres: the prefix sum of the counts rides in the fill pass (16 KB of LDS)

// (xcd_contiguous: common.h)
// MODE 0 = count: hits per centre, and the ballot of every 64-candidate round is saved (`masks`).
// MODE 1 = fill: writes the (j, i) rows at startIdx[i]. Windows whose rounds fit the saved masks are a pure
//          compaction -- no point is loaded and no distance evaluated a second time; larger windows are searched again.
// LEAN (foreground launches): the test loop carries no bounds -- the last round of a segment is padded with points out of
// anybody's reach --, is unrolled over the four rounds of a segment, and the count pass keeps their ballots in lanes 0..3
// and stores them with ONE instruction per (centre, segment). Faster alone (count pass on the room 35 -> 26 us), but its
// denser LDS-read / VALU bursts cost the convolution kernels it runs beside more than the search saves (pipelined step
// 0.640 -> 0.672 ms): background launches (mccnn_background_launches) keep the plain loop.
// (the kernel's body: workgroup `blk` of `nblk` -- the single launch, and one geometry's share of a batch launch)
template <int MODE, bool LEAN>
__device__ __forceinline__ void neigh_window_body(const int blk, const int nblk, const float* __restrict__ centres, const int* __restrict__ cb, int m,
                                                    const float* __restrict__ pts, const int* __restrict__ cells,
                                                    const float* __restrict__ mn, const float* __restrict__ mx, int B, int nc,
                                                    float radius, int scaleInv, const int* __restrict__ order,
                                                    int* __restrict__ cnt, unsigned long long* __restrict__ masks,
                                                    const int* __restrict__ startIdx, int* __restrict__ packed,
                                                    int capacity, unsigned long long* __restrict__ zeroWords, int numZero,
                                                    int G /* centres per wave, 1 .. 32 */, float Tabs,
                                                    const int* __restrict__ scanCnt, int* __restrict__ startOut,
                                                    int* __restrict__ totalDev, int* __restrict__ totalHost) {
    constexpr bool FILL = MODE == 1;
    // the status words of the prefix sum that follows the count pass (scan.hip): cleared here, no launch of their own
    if (!FILL && blk == 0)
        for (int k = threadIdx.x; k < numZero; k += blockDim.x) zeroWords[k] = 0ull;
    __shared__ float4 win[4][MCCNN_NW_CAP];
    __shared__ int2 ctab[4][32];
    extern __shared__ int scanLds[];   // [m] exclusive prefix of the counts (scan mode of the fill pass only)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (FILL && scanCnt) {
        // Lists of at most MCCNN_NW_SCAN_M centres (the coarse levels of a hierarchy): the prefix sum between the two passes
        // is two tiles at most -- every workgroup of the fill pass computes it for itself in LDS (16 KB of counts, a block scan)
        // instead of waiting for a launch of its own; workgroup 0 also writes startIdx and the edge total (device word and
        // the caller's pinned word).
        __shared__ int wtot[4];
        constexpr int PER = MCCNN_NW_SCAN_M / 256;
        int v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x * PER + k;
            v[k] = idx < m ? scanCnt[idx] : 0;
            sum += v[k];
        }
        const int incl = wave_incl_scan(sum);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w = 0; w < wave; ++w) run += wtot[w];
        const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x * PER + k;
            if (idx < m) {
                scanLds[idx] = run;
                if (blk == 0) startOut[idx] = run;
            }
            run += v[k];
        }
        if (blk == 0 && threadIdx.x == 0) {
            *totalDev = total;
            if (totalHost) *totalHost = total;
        }
        __syncthreads();
    }
    const int g0 = (xcd_contiguous(blk, nblk) * 4 + wave) * G;
    if (g0 >= m) return;
    float4* lw = win[wave];
    int2* tab = ctab[wave];
    // lanes 0..7: one centre each
    const int ci = g0 + lane;
    const bool own = lane < G && ci < m;
    const int i = own ? (order ? order[ci] : ci) : 0;
    CentreCtx c = centre_ctx(centres, cb, mn, mx, i, B, nc, radius, scaleInv, Tabs);
    const int key = own ? ((c.b * nc + c.x) * nc + c.y) * nc + c.z : -1;
    int count = 0;                                   // hits of this lane's centre so far
    const int base = (FILL && own) ? (scanCnt ? scanLds[i] : startIdx[i]) : 0;
    unsigned todo = (unsigned)(__ballot(own) & ((1ull << G) - 1));
    const int2* ct = reinterpret_cast<const int2*>(cells);
    int2* out = reinterpret_cast<int2*>(packed);
    while (todo) {
        const int lead = __builtin_ctz(todo);
        const int wkey = __builtin_amdgcn_readlane(key, lead);
        const unsigned members = (unsigned)(__ballot(own && key == wkey)) & todo;
        todo &= ~members;
        // the 27 cell ranges of the window: lane o < 27 owns table entry o
        const int wb = __builtin_amdgcn_readlane(c.b, lead), wx = __builtin_amdgcn_readlane(c.x, lead);
        const int wy = __builtin_amdgcn_readlane(c.y, lead), wz = __builtin_amdgcn_readlane(c.z, lead);
        int j0 = 0, len = 0;
        if (lane < 27) {
            const int slab = lane / 9, u = lane - slab * 9;
            const int X = wx + 1 - (u % 3), Y = wy + 1 - (u / 3), Z = wz + 1 - slab;
            if (X >= 0 && X < nc && Y >= 0 && Y < nc && Z >= 0 && Z < nc) {
                const int2 r = ct[(size_t)wb * nc * nc * nc + (size_t)X * nc * nc + (size_t)Y * nc + Z];
                j0 = r.x;
                len = r.y - r.x;
            }
        }
        const int off = wave_incl_scan(len) - len;   // flat offset of cell `lane` in the canonical candidate list
        const int total = __builtin_amdgcn_readlane(off + len, 26);
        __builtin_amdgcn_wave_barrier();
        if (lane < 27) tab[lane] = make_int2(j0 - off, off + len);
        __builtin_amdgcn_wave_barrier();
        // neighbour index of flat position f of the window: its cell is found by a 5-step binary search over the 27
        // cell end offsets (kept in LDS)
        auto flat_to_j = [&](int f) -> int {
            int lo = 0, hi = 26;  // smallest o with end[o] > f
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int mid = (lo + hi) >> 1;
                const bool right = tab[mid].y <= f;
                lo = right ? mid + 1 : lo;
                hi = right ? hi : mid;
            }
            return tab[lo].x + f;  // (j0 - off) + f
        };
        if (FILL && total <= MCCNN_NW_ROUNDS * 64) {
            // pure compaction from the saved ballots: lane = candidate of round r, the same for every centre of the cell
            int jr[MCCNN_NW_ROUNDS];
#pragma unroll
            for (int r = 0; r < MCCNN_NW_ROUNDS; ++r) jr[r] = (r * 64 < total) ? flat_to_j(min(r * 64 + lane, total - 1)) : 0;
            unsigned mem = members;
            while (mem) {
                const int cl = __builtin_ctz(mem);
                mem &= mem - 1;
                const int cbase = __builtin_amdgcn_readlane(base, cl), cid = __builtin_amdgcn_readlane(i, cl);
                const unsigned long long* mrow = masks + (size_t)(__builtin_amdgcn_readfirstlane(g0) + cl) * MCCNN_NW_ROUNDS;  // by visiting position (see the count pass); scalar address
                // all eight words of the centre at once (ONE 64-byte scalar load; words of rounds beyond the window are
                // never looked at): a load and a wait per round serialised ~5 scalar round trips per centre
                unsigned long long mw[MCCNN_NW_ROUNDS];
#pragma unroll
                for (int r = 0; r < MCCNN_NW_ROUNDS; ++r) mw[r] = mrow[r];
                // the hit lanes of a round come straight from the saved ballot (s_and_saveexec: no per-lane bit test); the
                // capacity is checked per centre, per lane only for the one centre that straddles it
                int hits = 0;
#pragma unroll
                for (int r = 0; r < MCCNN_NW_ROUNDS; ++r)
                    if (r * 64 < total) hits += __builtin_popcountll(mw[r]);
                const bool fits = cbase + hits <= capacity;  // capacity < E: see _fill
                int run = 0;
#pragma unroll
                for (int r = 0; r < MCCNN_NW_ROUNDS; ++r) {
                    if (r * 64 < total) {
                        const unsigned long long bm = mw[r];
#ifdef MCCNN_NW_NO_INVB
                        if ((bm >> lane) & 1ull) {
#else
                        if (__builtin_amdgcn_inverse_ballot_w64(bm)) {
#endif
                            const int pos = cbase + run + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                            if (fits) out[pos] = make_int2(jr[r], cid);
                            else if (pos < capacity) out[pos] = make_int2(jr[r], cid);
                        }
                        run += __builtin_popcountll(bm);
                    }
                }
                // (Round 5, measured and dropped -- 8 rooms, fill pass 167-170 us: the hits of a centre gathered in LDS and
                // written with ONE store per centre instead of one per round, double-buffered: 170 us, the pass is not bound
                // by store issue; on top of it the ballots of all centres of a wave loaded up front with coalesced vector
                // loads instead of a scalar load per centre: 162 us, but 30.6 instead of 28.1 us on one room; the pass with
                // the centres in index order, so that every 128-byte line of `packed` has one writer: 288 us -- without
                // the cell-coherent order every centre pays for its own window table, far more than the partial lines cost.)
                // (Measured and dropped: the cell of a flat position from a bit plane of cell ends -- ds_or per non-empty
                // cell, a population count and one v_mbcnt per round instead of the 5-step binary search: the plane's set-up
                // per window (two more wave barriers, an LDS atomic, a scan) eats what the searches cost, -1 % on one box.
                // MCCNN_NW_NO_INVB: the per-lane bit test instead of the inverse ballot, +2.7 % on the room. The mask words of
                // the NEXT centre requested while this one is compacted (and the first before the cell searches): +3 %.)
            }
            continue;
        }
        for (int seg = 0; seg < total; seg += MCCNN_NW_CAP) {
            const int segN = min(MCCNN_NW_CAP, total - seg);
            // stage [seg, seg + segN) of the flat list: lane = flat position, one load and one LDS write per position
            for (int r = 0; r < segN; r += 64) {
                const int j = flat_to_j(min(seg + r + lane, total - 1));
                const float* q = pts + (size_t)j * 3;  // 12-byte rows: one dwordx3 load
                if (LEAN) {
                    const bool real = r + lane < segN;
                    lw[r + lane] = make_float4(real ? q[0] : 3.0e18f, q[1], q[2], __int_as_float(j));
                } else if (r + lane < segN) {
                    lw[r + lane] = make_float4(q[0], q[1], q[2], __int_as_float(j));
                }
            }
            __builtin_amdgcn_wave_barrier();
            // every centre of this cell against the staged candidates
            unsigned mem = members;
            if constexpr (LEAN && !FILL) {
                // the count pass of foreground launches tests TWO centres per candidate: packed f32 arithmetic (8
                // v_pk_* + 2 compares per 64 candidates and centre pair instead of 2 x 9 instructions; the same
                // operations in the same order per half, so the same decisions) and one LDS read serves both
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                while (mem) {
                    const int c0 = __builtin_ctz(mem);
                    mem &= mem - 1;
                    const bool two = mem != 0;
                    const int c1 = two ? __builtin_ctz(mem) : c0;
                    mem &= mem - 1;   // (0 & anything stays 0)
                    const f32x2 CX = {readlane_f(c.cx, c0), readlane_f(c.cx, c1)};
                    const f32x2 CY = {readlane_f(c.cy, c0), readlane_f(c.cy, c1)};
                    const f32x2 CZ = {readlane_f(c.cz, c0), readlane_f(c.cz, c1)};
                    const float T0 = readlane_f(c.T, c0), T1 = readlane_f(c.T, c1);
                    int n0 = __builtin_amdgcn_readlane(count, c0), n1 = __builtin_amdgcn_readlane(count, c1);
                    unsigned lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
                    auto round = [&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        const float4 p = lw[K * 64 + lane];
                        const f32x2 X = {p.x, p.x}, Y = {p.y, p.y}, Z = {p.z, p.z};
                        const f32x2 dx = X - CX, dy = Y - CY, dz = Z - CZ;
                        const f32x2 d2 = dx * dx + dy * dy + dz * dz;   // point_dist2, both halves (no contraction: -ffp-contract=off)
                        const unsigned long long b0 = __ballot(d2.x < T0), b1 = __ballot(d2.y < T1);
                        writelane_u<K>(lo0, (unsigned)b0);
                        writelane_u<K>(hi0, (unsigned)(b0 >> 32));
                        writelane_u<K>(lo1, (unsigned)b1);
                        writelane_u<K>(hi1, (unsigned)(b1 >> 32));
                        n0 += __builtin_popcountll(b0);
                        n1 += __builtin_popcountll(b1);
                    };
                    static_assert(MCCNN_NW_CAP == 256, "four rounds per segment below");
                    round(std::integral_constant<int, 0>{});
                    if (segN > 64) round(std::integral_constant<int, 1>{});
                    if (segN > 128) round(std::integral_constant<int, 2>{});
                    if (segN > 192) round(std::integral_constant<int, 3>{});
                    const int round0 = seg >> 6;  // MCCNN_NW_CAP is a multiple of 64: a segment starts on a round
                    if (lane < ((segN + 63) >> 6) && round0 + lane < MCCNN_NW_ROUNDS) {
                        unsigned long long* row = masks + (size_t)__builtin_amdgcn_readfirstlane(g0) * MCCNN_NW_ROUNDS + round0 + lane;
                        row[(size_t)c0 * MCCNN_NW_ROUNDS] = ((unsigned long long)hi0 << 32) | lo0;
                        if (two) row[(size_t)c1 * MCCNN_NW_ROUNDS] = ((unsigned long long)hi1 << 32) | lo1;
                    }
                    if (lane == c0) count = n0;
                    if (two && lane == c1) count = n1;
                }
            } else
            while (mem) {
                const int cl = __builtin_ctz(mem);
                mem &= mem - 1;
                const float cx = readlane_f(c.cx, cl), cy = readlane_f(c.cy, cl);
                const float cz = readlane_f(c.cz, cl), T = readlane_f(c.T, cl);
                const int cid = __builtin_amdgcn_readlane(i, cl);
                int cbase = 0, ccount = __builtin_amdgcn_readlane(count, cl);
                if (FILL) cbase = __builtin_amdgcn_readlane(base, cl);
                // (Measured and withdrawn: this loop without bounds -- the last round padded with unreachable points -- unrolled
                // over the four rounds of a segment, the ballots kept in lanes 0..3 and stored once per segment: the count pass
                // alone 35 -> 26 us and a sequential step 0.740 -> 0.728 ms, but the PIPELINED step 0.640 -> 0.672 ms, whatever
                // the centres per wave, the store form or the issue priority of the convolution kernels (s_setprio): beside
                // the convolution kernels the denser LDS-read / VALU bursts cost them more than the search saves.)
                if (LEAN) {
                    unsigned mlo = 0, mhi = 0;
                    static_assert(MCCNN_NW_CAP % 64 == 0 && MCCNN_NW_CAP <= 512, "rounds of a segment: lanes 0..7 below");
                    auto round = [&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        const float4 p = lw[K * 64 + lane];
                        const bool hit = point_dist2(p.x, p.y, p.z, cx, cy, cz) < T;
                        const unsigned long long bm = __ballot(hit);
                        if (FILL && hit) {
                            const int pos = cbase + ccount + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                            if (pos < capacity) out[pos] = make_int2(__float_as_int(p.w), cid);  // capacity < E: see _fill
                        }
                        if (!FILL) {
                            writelane_u<K>(mlo, (unsigned)bm);
                            writelane_u<K>(mhi, (unsigned)(bm >> 32));
                        }
                        ccount += __builtin_popcountll(bm);
                    };
                    round(std::integral_constant<int, 0>{});
                    if (segN > 64) round(std::integral_constant<int, 1>{});
                    if (segN > 128) round(std::integral_constant<int, 2>{});
                    if (segN > 192) round(std::integral_constant<int, 3>{});
                    if constexpr (MCCNN_NW_CAP > 256) {
                        if (segN > 256) round(std::integral_constant<int, 4>{});
                        if (segN > 320) round(std::integral_constant<int, 5>{});
                    }
                    if constexpr (MCCNN_NW_CAP > 384) {
                        if (segN > 384) round(std::integral_constant<int, 6>{});
                        if (segN > 448) round(std::integral_constant<int, 7>{});
                    }
                    if (!FILL) {
                        const int round0 = seg >> 6;  // MCCNN_NW_CAP is a multiple of 64: a segment starts on a round
                        if (lane < ((segN + 63) >> 6) && round0 + lane < MCCNN_NW_ROUNDS)
                            masks[(size_t)(__builtin_amdgcn_readfirstlane(g0) + cl) * MCCNN_NW_ROUNDS + round0 + lane] =
                                ((unsigned long long)mhi << 32) | mlo;
                    }
                } else
                for (int r = 0; r < segN; r += 64) {
                    const int t = r + lane;
                    const float4 p = lw[min(t, segN - 1)];
                    const bool hit = t < segN && point_dist2(p.x, p.y, p.z, cx, cy, cz) < T;
                    const unsigned long long bm = __ballot(hit);
                    if (FILL && hit) {
                        const int pos = cbase + ccount + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                        if (pos < capacity) out[pos] = make_int2(__float_as_int(p.w), cid);  // capacity < E: see _fill
                    }
                    if (!FILL) {
                        const int round = (seg + r) >> 6;
                        // indexed by the centre's POSITION in the visiting order, not by its id: the 8 centres of a wave own
                        // 512 contiguous bytes (4 lines) in both passes; by id they were 8 scattered 64-byte rows. (One plane
                        // of m words per round cut the counter traffic further, 90.8 -> 78.1 MB on the room, but made the
                        // fill read a line per round and centre: 36.7 instead of 27.1 us.)
                        if (round < MCCNN_NW_ROUNDS && lane == 0) masks[(size_t)(__builtin_amdgcn_readfirstlane(g0) + cl) * MCCNN_NW_ROUNDS + round] = bm;
                    }
                    ccount += __builtin_popcountll(bm);
                }
                if (lane == cl) count = ccount;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!FILL && own) cnt[i] = count;
}

template <int MODE, bool LEAN>
__global__ __launch_bounds__(256) void neigh_window(const float* __restrict__ centres, const int* __restrict__ cb, int m,
                                                    const float* __restrict__ pts, const int* __restrict__ cells,
                                                    const float* __restrict__ mn, const float* __restrict__ mx, int B, int nc,
                                                    float radius, int scaleInv, const int* __restrict__ order,
                                                    int* __restrict__ cnt, unsigned long long* __restrict__ masks,
                                                    const int* __restrict__ startIdx, int* __restrict__ packed,
                                                    int capacity, unsigned long long* __restrict__ zeroWords, int numZero,
                                                    int G /* centres per wave, 1 .. 32 */, float Tabs,
                                                    const int* __restrict__ scanCnt, int* __restrict__ startOut,
                                                    int* __restrict__ totalDev, int* __restrict__ totalHost) {
    neigh_window_body<MODE, LEAN>((int)blockIdx.x, (int)gridDim.x, centres, cb, m, pts, cells, mn, mx, B, nc, radius, scaleInv, order, cnt,
                                  masks, startIdx, packed, capacity, zeroWords, numZero, G, Tabs, scanCnt, startOut, totalDev, totalHost);
}

// One launch for the count (or the fill) pass of a BATCH of searches (mccnn_geometry_build_batch): the plain loop of the
// background launches, the prefix sum of the counts a launch of its own in between (scan.hip's batch form).
template <int MODE>
__global__ __launch_bounds__(256) void neigh_window_batch(NeighBatch nbt, BatchBlocks bb) {
    int local, blocks;
    const NeighItem& g = nbt.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    neigh_window_body<MODE, false>(local, blocks, g.centres, g.cb, g.m, g.pts, g.cells, g.mn, g.mx, g.B, g.nc, g.radius, g.scaleInv, g.order,
                                   g.cnt, g.masks, g.startIdx, g.packed, g.capacity, g.zeroWords, g.numZero, g.G, g.Tabs, nullptr, nullptr,
                                   nullptr, nullptr);
}

__global__ __launch_bounds__(256) void invert_perm_k(const int* __restrict__ newIdx, int n, int* __restrict__ inv) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[newIdx[i]] = i;
}

// ------------------------------------------------------------------ compute_pdf.cu:40-94
// Mode 0: one thread per edge (j, i), the reference's arithmetic (3 double-precision exps per pair).
__global__ __launch_bounds__(256) void pdf_edges_ref(const float* __restrict__ pts, const int* __restrict__ bids,
                                                 const int* __restrict__ startIdx, int m,
                                                 const int2* __restrict__ packed, int e, const float* __restrict__ mn,
                                                 const float* __restrict__ mx, int B, float window, float radius,
                                                 int scaleInv, float* __restrict__ pdfs) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= e) return;
    int2 pr = packed[t];
    int cur = pr.x, centre = pr.y;
    float cx = pts[(size_t)cur * 3], cy = pts[(size_t)cur * 3 + 1], cz = pts[(size_t)cur * 3 + 2];
    int b = clamp_batch(bids[cur], B);
    float ext = max_extent(mn, mx, b);
    float R = scaleInv ? radius * ext : radius;
    int i0 = startIdx[centre];
    int i1 = (centre < m - 1) ? startIdx[centre + 1] : e;
    const float h = window;
    const float invH = 1 / h;
    const float invRadH = (float)(1.0 / (double)(R * h));  // compute_pdf.cu:74
    float pdf = 0.0f;
    for (int it = i0; it < i1; ++it) {
        size_t q = (size_t)packed[it].x * 3;
        float d0 = (pts[q] - cx) * invRadH;
        float d1 = (pts[q + 1] - cy) * invRadH;
        float d2 = (pts[q + 2] - cz) * invRadH;
        // compute_pdf.cu:85-88, double sub-expressions rounded to float per statement
        float g = (float)((double)invH * ((0.39894228) * exp((-0.5) * (double)d0 * (double)d0)));
        g = (float)((double)(g * invH) * ((0.39894228) * exp((-0.5) * (double)d1 * (double)d1)));
        g = (float)((double)(g * invH) * ((0.39894228) * exp((-0.5) * (double)d2 * (double)d2)));
        pdf += g;
    }
    pdfs[t] = pdf / ((float)i1 - i0);  // compute_pdf.cu:92
}

// ---- single-precision KDE (mode 1) ------------------------------------------------------------------
// Pre-pass: one float4 per edge with the neighbour's RAW coordinates and, in .w, the scale 1/(R_b h) of its cloud.
// The pair loop subtracts raw coordinates -- the difference of two nearby floats is (nearly) exact, as in the
// reference's (p_a - p_b) * invRadH (compute_pdf.cu:78-80) -- and applies the scale to the squared distance; scaling
// the absolute coordinates first would lose |p| / (R h) * 2^-24 to cancellation (1e-3 for a scene 100 m from the
// origin at r = 0.1).
__global__ __launch_bounds__(256) void pdf_edge_coords(const float* __restrict__ pts, const int* __restrict__ bids,
                                                       const int2* __restrict__ packed, int e,
                                                       const float* __restrict__ mn, const float* __restrict__ mx,
                                                       int B, float window, float radius, int scaleInv,
                                                       float4* __restrict__ sc, const int* __restrict__ eDev) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (eDev) e = min(e, max(*eDev, 0));  // device-side edge count (e is then the capacity of the lists)
    if (t >= e) return;
    int j = packed[t].x;
    float R = scaleInv ? radius * max_extent(mn, mx, clamp_batch(bids[j], B)) : radius;
    float s = (float)(1.0 / (double)(R * window));
    const float* p = pts + (size_t)j * 3;
    sc[t] = make_float4(p[0], p[1], p[2], s);
}

// One row of the KDE with the row's points (a) in the lanes and its points (b) wave-uniform: scalar loads into SGPRs,
// no vector memory or LDS traffic in the pair loop, no divergence between rows of different length (a thread per edge
// walks its row with one gather per pair and idles while a longer row in the same wave finishes: 136 -> 100 us on the
// 100k room). Same arithmetic and summation order as the thread-per-edge form it replaced: identical results. (Reading
// the pre-scaled coordinates per POINT through the neighbour index instead of the per-edge copy makes the scalar loads
// dependent: measured slower, 124 us.)
__device__ __forceinline__ void pdf_row_scalar(const float4* __restrict__ rowp, int k, int i0, int i1, int lane, float norm,
                                               float* __restrict__ pdfs) {
    for (int a0 = 0; a0 < k; a0 += 64) {
        const int a = a0 + lane;
        const float4 me = rowp[min(a, k - 1)];
        const float c = (-0.5f * 1.44269504088896f) * (me.w * me.w);  // exp(-|d s|^2 / 2) = exp2(c |d|^2), s = 1 / (R h)
        float acc = 0.f;
        int b = 0;
        for (; b + 4 <= k; b += 4) {
            const float4 q0 = rowp[b], q1 = rowp[b + 1], q2 = rowp[b + 2], q3 = rowp[b + 3];
            float dx, dy, dz;
            dx = q0.x - me.x; dy = q0.y - me.y; dz = q0.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q1.x - me.x; dy = q1.y - me.y; dz = q1.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q2.x - me.x; dy = q2.y - me.y; dz = q2.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q3.x - me.x; dy = q3.y - me.y; dz = q3.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
        for (; b < k; ++b) {
            const float4 q0 = rowp[b];
            const float dx = q0.x - me.x, dy = q0.y - me.y, dz = q0.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
        if (a < k) pdfs[i0 + a] = (acc * norm) / ((float)i1 - i0);
    }
}

// Mode 2, row form on the VALU only: one wave per centre.
__global__ __launch_bounds__(256) void pdf_rows(const float4* __restrict__ sc, const int* __restrict__ startIdx, int m,
                                                int e, float window, float* __restrict__ pdfs,
                                                const int* __restrict__ eDev) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int cap = e;
    if (eDev) e = min(e, max(*eDev, 0));
    const int lane = threadIdx.x & 63;
    const int i0 = __builtin_amdgcn_readfirstlane(startIdx[row]);
    int i1 = __builtin_amdgcn_readfirstlane((row < m - 1) ? startIdx[row + 1] : e);
    if (eDev) i1 = min(i1, cap);  // a capacity below the true total: rows are cut, the caller repeats with the exact size
    const int k = i1 - i0;
    if (k <= 0) return;
    const float invH = 1.0f / window;
    const float g1 = invH * 0.39894228f;
    pdf_row_scalar(sc + i0, k, i0, i1, lane, g1 * g1 * g1, pdfs);
}

// Mode 1, the pair sums as a Gram matrix on the matrix cores: one wave per centre, v_mfma_f32_16x16x4_f32 per 16 x 16
// tile of pairs. With u = (p - o) s sqrt(log2(e) / 2), o = the row's first point (every point of a row lies within 2 R of
// it, so |u| <= 2 / h and the subtraction p - o of nearby floats is (nearly) exact -- a scene 500 m from the origin costs
// nothing) and q = |u|^2:
//     |u_i - u_j|^2 = [u_i, 1] . [-2 u_j, q_j] + q_i      (K = 4: one instruction, q_i rides in as the C operand)
// and the weight is exp2(-that): per PAIR one v_exp_f32 and one add on the VALU instead of the nine instructions of the
// scalar-broadcast loop, and a row of 45 points fills 88 % of its 3 x 3 tiles where it filled 70 % of 64 lanes. Error
// against the subtract-first form: <= 1e-5 relative per value (|u|^2 <= 300, measured 9e-6 worst over rows up to 120
// points; tests hold 1e-4 against the oracle). Operands are staged per wave in LDS as planes ([ux uy uz 1] for A,
// [-2ux -2uy -2uz q] for B; rows of the last tile beyond k carry q = 3e38, whose weights underflow to exactly 0).
// Rows longer than MCCNN_PDF_CAP points take the scalar loop.
#ifndef MCCNN_PDF_CAP
#define MCCNN_PDF_CAP 192
#endif
#ifndef MCCNN_PDF_ROWS
#define MCCNN_PDF_ROWS 4
#endif
// MCCNN_PDF_ROWS // consecutive rows per wave: the next row's points are requested while this row's tiles run
typedef float pdf_v4f __attribute__((ext_vector_type(4)));
#if defined(MCCNN_PDF_ABL) && MCCNN_PDF_ABL == 1   // timing ablations only (wrong results): 1 no exponentials, 2 no tiles, 3 no MFMA
#define PDF_EXP(x) (x)
#else
#define PDF_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
#if defined(MCCNN_PDF_ABL) && MCCNN_PDF_ABL == 3
#define PDF_MFMA(a, b, c) ((c) * (a) + (b))
#else
#define PDF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif
struct PdfPoint { float x, y, z; };
__device__ __forceinline__ PdfPoint pdf_gather(const float* __restrict__ pts, int j) {
    const float* p = pts + (size_t)j * 3;
    return PdfPoint{p[0], p[1], p[2]};
}

// Row longer than the tile planes hold: subtract-first pair loop (the arithmetic of pdf_row_scalar) with the row staged
// through the wave's LDS planes MCCNN_PDF_CAP points at a time (one broadcast ds_read_b128 per pair step).
// aBegin / aStep: the row's blocks of 64 output values are shared out over the waves of the workgroup (a row of k points
// costs k^2 pair terms: walked by ONE wave, the longest row of a pooling list -- 6 344 centres with 218 neighbours on
// average, some with > 1 000 -- alone set the kernel's time: 597 us on BASELINE cfg2 Pool_1).
// (force-inlined: through a non-inlined call the LDS planes arrive as a generic pointer and every pair step became a
// FLAT load with a full wait -- 0.58 T pair terms/s on BASELINE cfg2 Pool_1 against 3.2 T on the tile path)
__device__ __forceinline__ void pdf_row_long(const float* __restrict__ pts, const int2* __restrict__ rowpk, int k, int rowStart,
                                          int lane, float s, float scale, float* __restrict__ P,
                                          float* __restrict__ pdfs, int aBegin, int aStep) {
    float4* __restrict__ P4 = reinterpret_cast<float4*>(P);
    const float cc = (-0.5f * 1.44269504088896f) * (s * s);
    for (int a0 = aBegin; a0 < k; a0 += aStep) {
        const int a = a0 + lane;
        const PdfPoint me = pdf_gather(pts, rowpk[min(a, k - 1)].x);
        float acc = 0.f;
        for (int b0 = 0; b0 < k; b0 += MCCNN_PDF_CAP) {
            const int nb = min(MCCNN_PDF_CAP, k - b0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < nb; t += 64) {
                const PdfPoint q = pdf_gather(pts, rowpk[b0 + t].x);
                P4[t] = make_float4(q.x, q.y, q.z, 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
            for (int t = 0; t < nb; ++t) {
                const float4 q = P4[t];
                const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
                acc += __builtin_amdgcn_exp2f(cc * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            }
        }
        if (a < k) pdfs[rowStart + a] = acc * scale;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct PdfLongRow { int rowStart, k; float s, scale; };
template <int WAVES>  // 4: lists of many rows (4 rows per wave); 16: lists of few rows (one row per wave)
__device__ __forceinline__ void pdf_rows_mfma_body(const int blk, const float* __restrict__ pts, const int* __restrict__ bids,
                                                     const int2* __restrict__ packed, const int* __restrict__ startIdx,
                                                     int m, int e, const float* __restrict__ mn,
                                                     const float* __restrict__ mx, int B, float window, float radius,
                                                     int scaleInv, float* __restrict__ pdfs,
                                                     const int* __restrict__ eDev, int rowsPerWave) {
    __shared__ __attribute__((aligned(16))) float planes[WAVES][5 * MCCNN_PDF_CAP];  // per wave: ux, uy, uz, q, ones
    // rows longer than the planes: set aside here and walked by ALL waves of the workgroup afterwards (pdf_row_long)
    __shared__ PdfLongRow longRows[WAVES * MCCNN_PDF_ROWS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave-uniform: rows, tile counts and loops live in SGPRs
    __shared__ int anyLong;
    for (int t = threadIdx.x; t < WAVES * MCCNN_PDF_ROWS; t += WAVES * 64) longRows[t].k = 0;
    if (threadIdx.x == 0) anyLong = 0;
    __syncthreads();
    const int r0 = (blk * WAVES + wave) * rowsPerWave;
    float* __restrict__ P = planes[wave];
    const int c = lane >> 4, mm = lane & 15;
    const float bscale = (c == 3) ? 1.0f : -2.0f;
    if (r0 < m) {
    const int nr = min(rowsPerWave, m - r0);
    const int cap = e;
    if (eDev) e = min(e, max(*eDev, 0));
    // row bounds of this wave's rows in lanes 0..nr (a capacity below the true total cuts rows: the caller repeats)
    int stv = e;
    if (lane <= nr && r0 + lane < m) stv = startIdx[r0 + lane];
    stv = min(stv, cap);
    const float invH = 1.0f / window;
    const float g1 = invH * 0.39894228f;
    const float norm = g1 * g1 * g1;
    const float* __restrict__ pb = P + c * MCCNN_PDF_CAP + mm;                   // B: component c of point 16 t + mm (c = 3: q)
    const float* __restrict__ pa = P + (c == 3 ? 4 : c) * MCCNN_PDF_CAP + mm;    // A: the same, with 1 in place of q
    const float* __restrict__ pq = P + 3 * MCCNN_PDF_CAP + 4 * c;                // C: q of points 16 t + 4 c + (0..3)
    for (int t = lane; t < MCCNN_PDF_CAP; t += 64) P[4 * MCCNN_PDF_CAP + t] = 1.0f;
    // 1 / (R h): one value for all clouds with an absolute radius, per row otherwise
    float sAbs = 0.f;
    if (!scaleInv) sAbs = (float)(1.0 / (double)(radius * window));

    // Two loads deep: the neighbour INDICES of row r + 1 and the POINTS of row r are requested one row ahead (the
    // first 64 entries of a row; longer rows fetch the rest when they are staged).
    // (An empty row requests nothing: with a device-side count of 0 not even packed[0] is a valid index.)
    int i0 = __builtin_amdgcn_readlane(stv, 0), i1 = __builtin_amdgcn_readlane(stv, 1);
    int jn = 0, bn = 0;
    PdfPoint pn{0.f, 0.f, 0.f};
    if (i1 > i0) {
        jn = packed[min(i0 + lane, i1 - 1)].x;
        pn = pdf_gather(pts, jn);
        bn = bids[jn];
    }
    if (nr > 1) {
        const int n1 = __builtin_amdgcn_readlane(stv, 2);
        if (n1 > i1) jn = packed[min(i1 + lane, n1 - 1)].x;
    }
    for (int rr = 0; rr < nr; ++rr) {
        const PdfPoint me = pn;
        const int bme = bn;
        const int k = i1 - i0, rowStart = i0;
        if (rr + 1 < nr) {
            i0 = i1;
            i1 = __builtin_amdgcn_readlane(stv, rr + 2);
            if (i1 > i0) {
                pn = pdf_gather(pts, jn);
                bn = bids[jn];
            }
            if (rr + 2 < nr) {
                const int n1 = __builtin_amdgcn_readlane(stv, rr + 3);
                if (n1 > i1) jn = packed[min(i1 + lane, n1 - 1)].x;
            }
        }
        if (k <= 0) continue;
        const int2* __restrict__ rowpk = packed + rowStart;
        // 1 / (R h) of the row's cloud (every neighbour of a centre lies in the centre's cloud)
        float s = sAbs;
        if (scaleInv) {
            const int b0 = clamp_batch(__builtin_amdgcn_readfirstlane(bme), B);
            s = (float)(1.0 / (double)((radius * max_extent(mn, mx, b0)) * window));
        }
        const float scale = norm * __builtin_amdgcn_rcpf((float)k);  // 1 ulp: this mode is not the bit-exact one
        if (k > MCCNN_PDF_CAP) {
            if (lane == 0) { longRows[wave * MCCNN_PDF_ROWS + rr] = PdfLongRow{rowStart, k, s, scale}; anyLong = 1; }
            continue;
        }
#if defined(MCCNN_PDF_ABL) && MCCNN_PDF_ABL == 2
        const int T = 0;
#else
        const int T = (k + 15) >> 4;
#endif
        const float ox = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, me.x)));
        const float oy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, me.y)));
        const float oz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, me.z)));
        const float sp = s * 0.84932180f;  // s sqrt(log2(e) / 2)
        for (int a0 = 0; a0 < 16 * T; a0 += 64) {
            const int a = a0 + lane;
            if (a < 16 * T) {
                const PdfPoint p = (a0 == 0) ? me : pdf_gather(pts, rowpk[min(a, k - 1)].x);
                const float ux = (p.x - ox) * sp, uy = (p.y - oy) * sp, uz = (p.z - oz) * sp;
                P[a] = ux;
                P[MCCNN_PDF_CAP + a] = uy;
                P[2 * MCCNN_PDF_CAP + a] = uz;
                P[3 * MCCNN_PDF_CAP + a] = (a < k) ? fmaf(uz, uz, fmaf(uy, uy, ux * ux)) : 3.0e38f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int J = 0; J < T; ++J) {
            const float bop = pb[16 * J] * bscale;
            float acc0 = 0.f, acc1 = 0.f;
            // tile pairs at compile-time LDS offsets (no address arithmetic in the loop); T <= MCCNN_PDF_CAP / 16
#pragma unroll
            for (int I = 0; I < MCCNN_PDF_CAP / 16; I += 2) {
                if (I + 2 > T) break;
                const float a0 = pa[16 * I], a1 = pa[16 * I + 16];
                const pdf_v4f c0 = *reinterpret_cast<const pdf_v4f*>(pq + 16 * I);
                const pdf_v4f c1 = *reinterpret_cast<const pdf_v4f*>(pq + 16 * I + 16);
                const pdf_v4f d0 = PDF_MFMA(a0, bop, c0);
                const pdf_v4f d1 = PDF_MFMA(a1, bop, c1);
                acc0 += (PDF_EXP(-d0[0]) + PDF_EXP(-d0[1])) + (PDF_EXP(-d0[2]) + PDF_EXP(-d0[3]));
                acc1 += (PDF_EXP(-d1[0]) + PDF_EXP(-d1[1])) + (PDF_EXP(-d1[2]) + PDF_EXP(-d1[3]));
            }
            if (T & 1) {
                const int I = T - 1;
                const float a0 = pa[16 * I];
                const pdf_v4f c0 = *reinterpret_cast<const pdf_v4f*>(pq + 16 * I);
                const pdf_v4f d0 = PDF_MFMA(a0, bop, c0);
                acc0 += (PDF_EXP(-d0[0]) + PDF_EXP(-d0[1])) + (PDF_EXP(-d0[2]) + PDF_EXP(-d0[3]));
            }
            // the four 16-lane groups hold the partial sums of four different quarter-sets of rows i: add them up with the
            // two gfx950 row-swap instructions (VALU; a ds_bpermute pair is two LDS round trips per column block)
            float acc = acc0 + acc1;
            {   // (inline asm: with both operands the same value the builtin hands back the first result twice)
                float lo = acc, hi = acc;
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                acc = lo + hi;  // rows 0,1: x0 + x1; rows 2,3: x2 + x3
                lo = acc;
                hi = acc;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                acc = lo + hi;
            }
            const int a = 16 * J + mm;
            if (c == 0 && a < k) pdfs[rowStart + a] = acc * scale;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }  // r0 < m
    __syncthreads();
    // A long row that fits the POOLED planes of the workgroup (WAVES x MCCNN_PDF_CAP points: 3 072 with 16 waves) is
    // staged once by all threads and its 16 x 16 tiles are shared out by column block: the same Gram-matrix arithmetic
    // as the short rows, at the matrix cores' rate (the subtract-first loop below runs at a third of it).
    if (!anyLong) return;
    constexpr int LC = WAVES * MCCNN_PDF_CAP;
    float* __restrict__ pool = &planes[0][0];
    int item = 0;
    for (int t = 0; t < WAVES * MCCNN_PDF_ROWS; ++t) {
        const PdfLongRow lr = longRows[t];
        if (lr.k <= 0) continue;
        if (lr.k <= LC) {
            const int k = lr.k, T = (k + 15) >> 4;
            const int2* __restrict__ rowpk = packed + lr.rowStart;
            const PdfPoint o = pdf_gather(pts, rowpk[0].x);  // the row's first point: every difference is within 2 R of it
            const float sp = lr.s * 0.84932180f;
            __syncthreads();  // the planes are free (the waves' own rows, the previous long row)
            for (int a = threadIdx.x; a < 16 * T; a += WAVES * 64) {
                const PdfPoint q = pdf_gather(pts, rowpk[min(a, k - 1)].x);
                const float ux = (q.x - o.x) * sp, uy = (q.y - o.y) * sp, uz = (q.z - o.z) * sp;
                pool[a] = ux;
                pool[LC + a] = uy;
                pool[2 * LC + a] = uz;
                pool[3 * LC + a] = (a < k) ? fmaf(uz, uz, fmaf(uy, uy, ux * ux)) : 3.0e38f;
                pool[4 * LC + a] = 1.0f;
            }
            __syncthreads();
            const float* __restrict__ pbL = pool + c * LC + mm;
            const float* __restrict__ paL = pool + (c == 3 ? 4 : c) * LC + mm;
            const float* __restrict__ pqL = pool + 3 * LC + 4 * c;
            for (int J = wave; J < T; J += WAVES) {
                const float bop = pbL[16 * J] * bscale;
                float acc0 = 0.f, acc1 = 0.f;
                int I = 0;
                for (; I + 2 <= T; I += 2) {
                    const float a0 = paL[16 * I], a1 = paL[16 * I + 16];
                    const pdf_v4f c0 = *reinterpret_cast<const pdf_v4f*>(pqL + 16 * I);
                    const pdf_v4f c1 = *reinterpret_cast<const pdf_v4f*>(pqL + 16 * I + 16);
                    const pdf_v4f d0 = PDF_MFMA(a0, bop, c0);
                    const pdf_v4f d1 = PDF_MFMA(a1, bop, c1);
                    acc0 += (PDF_EXP(-d0[0]) + PDF_EXP(-d0[1])) + (PDF_EXP(-d0[2]) + PDF_EXP(-d0[3]));
                    acc1 += (PDF_EXP(-d1[0]) + PDF_EXP(-d1[1])) + (PDF_EXP(-d1[2]) + PDF_EXP(-d1[3]));
                }
                if (T & 1) {
                    const float a0 = paL[16 * I];
                    const pdf_v4f c0 = *reinterpret_cast<const pdf_v4f*>(pqL + 16 * I);
                    const pdf_v4f d0 = PDF_MFMA(a0, bop, c0);
                    acc0 += (PDF_EXP(-d0[0]) + PDF_EXP(-d0[1])) + (PDF_EXP(-d0[2]) + PDF_EXP(-d0[3]));
                }
                float acc = acc0 + acc1;
                {
                    float lo = acc, hi = acc;
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                    acc = lo + hi;
                    lo = acc;
                    hi = acc;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                    acc = lo + hi;
                }
                const int a = 16 * J + mm;
                if (c == 0 && a < k) pdfs[lr.rowStart + a] = acc * lr.scale;
            }
            continue;
        }
        // beyond the pooled planes: work items = (row, block of 64 output values) dealt round-robin to the waves, the
        // subtract-first loop with the row streamed through each wave's own planes
        __syncthreads();
        const int blocks = (lr.k + 63) >> 6;
        for (int b = (wave - item % WAVES + WAVES) % WAVES; b < blocks; b += WAVES)
            pdf_row_long(pts, packed + lr.rowStart, lr.k, lr.rowStart, lane, lr.s, lr.scale, P, pdfs, b * 64, 1 << 30);
        item += blocks;
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pdf_rows_mfma(const float* __restrict__ pts, const int* __restrict__ bids,
                                                     const int2* __restrict__ packed, const int* __restrict__ startIdx,
                                                     int m, int e, const float* __restrict__ mn,
                                                     const float* __restrict__ mx, int B, float window, float radius,
                                                     int scaleInv, float* __restrict__ pdfs,
                                                     const int* __restrict__ eDev, int rowsPerWave) {
    pdf_rows_mfma_body<WAVES>((int)blockIdx.x, pts, bids, packed, startIdx, m, e, mn, mx, B, window, radius, scaleInv, pdfs, eDev, rowsPerWave);
}

// One launch for the KDE of a BATCH of lists (mccnn_geometry_build_batch): 16 waves per workgroup for every list (the
// pooled planes of long rows), rows per wave by the list's size.
__global__ __launch_bounds__(1024) void pdf_rows_mfma_batch(PdfBatch pb, BatchBlocks bb) {
    int local, blocks;
    const PdfItem& g = pb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    pdf_rows_mfma_body<16>(local, g.pts, g.bids, g.packed, g.startIdx, g.m, g.e, g.mn, g.mx, g.B, g.window, g.radius, g.scaleInv, g.pdfs,
                           g.eDev, g.rowsPerWave);
}

}  // namespace mccnn

using namespace mccnn;

extern "C" {

size_t mccnn_find_neighbors_workspace_bytes(int m, int n) {
    size_t m1 = (size_t)(m > 0 ? m : 1);
    (void)n;
    return align_up(m1 * 4) + scan_workspace_bytes((int)m1) + align_up(m1 * MCCNN_NW_ROUNDS * sizeof(unsigned long long)) + 256;
}

// Centres per wave: 8 consecutive centres of the visiting order share most of their windows on a large list (~8 points
// per cell), but a list with few centres needs the waves -- 6 344 pooling centres with ~900 candidates each ran 139 us per
// pass on 793 waves (BASELINE cfg2 Pool_1), the coarse levels of a hierarchy 20 us on a handful.
// sqrt_threshold (common.h) on the host: the same float operations, both square roots correctly rounded
static float sqrt_threshold_host(float R) {
    auto prev = [](float v) { uint32_t u; memcpy(&u, &v, 4); --u; memcpy(&v, &u, 4); return v; };
    auto next = [](float v) { uint32_t u; memcpy(&u, &v, 4); ++u; memcpy(&v, &u, 4); return v; };
    float t = R * R;
    for (int it = 0; it < 8 && t > 0.0f && sqrtf(prev(t)) >= R; ++it) t = prev(t);
    for (int it = 0; it < 8 && sqrtf(t) < R; ++it) t = next(t);
    return t;
}

// Background launches (mccnn_background_launches: the geometry of the next batch on a side queue, under the convolution
// kernels of the current one): the search kernels ask for 36 KB of LDS they do not use, which leaves 2-3 of their
// workgroups per CU instead of 8. Beside them the convolution waves of the same SIMD wait on LDS weight reads and MFMA
// hand-overs, and fewer neighbour waves competing for the LDS and VALU ports help them more than a slower search costs:
// pipelined 1to64 step on the room 0.636 -> 0.621 ms (10 / 16 KB: no gain; 60 KB: the search becomes the critical chain,
// 0.66 ms). The same limit on the KDE kernel LOSES (0.65 ms); alone the padded search is slower (sequential step 0.736 ->
// 0.762 ms), hence only for background launches. MCCNN_NW_LDS_PAD overrides the amount (A/B). Re-tuned after the forward
// pass of the headline layer got shorter (four edges per lane) and the background scans went to the two-launch form: 20 / 24 /
// 28 / 32 / 36 KB read 0.605-0.634 / 0.597 / 0.599-0.614 / 0.604 / 0.605 ms per pipelined step -> 24 KB.
static size_t neigh_lds_pad() {
    static const int forced = debug_int("nw_lds_pad", -1);
    if (forced >= 0) return (size_t)forced;
    return g_background ? 24000 : 0;
}
static bool neigh_lean() {
    static const int forced = debug_int("nw_lean", -1);  // A/B switch, read once
    if (forced >= 0) return forced != 0;
    return !g_background;
}
static int neigh_group(int m) {
    static const int forced = debug_int("nw_group", 0);  // A/B switch, read once
    if (forced >= 1 && forced <= 32) return forced;
    // centres per wave: more of them share a window's staging and the per-wave set-up (8 rooms, 800 k centres: 0.309 ms at
    // 8, 0.274 at 16, 0.269 at 24) -- as long as the launch still fills the chip (100 k centres: 0.0588 / 0.0580 / 0.0650)
    // (background launches keep 8: beside the convolution kernels the pipelined step of the room reads 0.596 ms at 8, 0.602-0.610 at 16)
    if (g_background) return m >= 32768 ? MCCNN_NW_G : (m >= 16384 ? 4 : (m >= 8192 ? 2 : 1));
    return m >= 400000 ? 24 : (m >= 65536 ? 16 : (m >= 32768 ? MCCNN_NW_G : (m >= 16384 ? 4 : (m >= 8192 ? 2 : 1))));
}

struct NeighWs {
    int* cnt;  // hits per centre
    void* scanws;
    unsigned long long* masks;  // per centre: the ballots of its first MCCNN_NW_ROUNDS candidate rounds (count -> fill)
};
static bool neigh_ws(void* ws, size_t ws_bytes, int m, int n, NeighWs& w) {
    if (!ws || ws_bytes < mccnn_find_neighbors_workspace_bytes(m, n)) return false;
    size_t m1 = (size_t)(m > 0 ? m : 1);
    Arena a(ws, ws_bytes);
    w.cnt = a.take<int>(m1);
    w.scanws = a.take<char>(scan_workspace_bytes((int)m1));
    w.masks = a.take<unsigned long long>(m1 * MCCNN_NW_ROUNDS);
    (void)n;
    return w.cnt && w.scanws && w.masks;
}

static int find_neighbors_count_impl(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                               int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                               int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                               int* start_idx, int* total_dev, int* total_host, void* ws, size_t ws_bytes, mccnn_stream_t stream,
                               bool skip_scan = false);

int mccnn_find_neighbors_count(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                               int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                               int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                               int* start_idx, int* total_dev, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    return find_neighbors_count_impl(centres, centre_batch_ids, m, sorted_pts, n, cell_indexs, aabb_min, aabb_max, batch_size,
                                     num_cells, radius, scale_inv, centre_order, start_idx, total_dev, nullptr, ws, ws_bytes, stream);
}

int mccnn_find_neighbors_count2(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                                int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                                int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                                int* start_idx, int* total_dev, int* total_host, void* ws, size_t ws_bytes,
                                mccnn_stream_t stream) {
    return find_neighbors_count_impl(centres, centre_batch_ids, m, sorted_pts, n, cell_indexs, aabb_min, aabb_max, batch_size,
                                     num_cells, radius, scale_inv, centre_order, start_idx, total_dev, total_host, ws, ws_bytes,
                                     stream);
}

static int find_neighbors_count_impl(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                               int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                               int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                               int* start_idx, int* total_dev, int* total_host, void* ws, size_t ws_bytes, mccnn_stream_t stream,
                               bool skip_scan) {
    if (m < 0 || n < 0 || batch_size <= 0 || num_cells <= 0 || !(radius > 0.0f) || !total_dev) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        int rc = launch_zero_words(total_dev, 1, s);
        if (!rc && total_host) rc = launch_zero_words(total_host, 1, s);   // (a pinned word the device can write)
        return rc;
    }
    if (!centres || !centre_batch_ids || !cell_indexs || !aabb_min || !aabb_max || !start_idx || (n > 0 && !sorted_pts))
        return MCCNN_E_BADARG;
    NeighWs w;
    if (!neigh_ws(ws, ws_bytes, m, n, w)) return MCCNN_E_WORKSPACE;
    const int G = neigh_group(m);
    const float Tabs = scale_inv ? 0.0f : sqrt_threshold_host(radius);
    unsigned long long* zw = (unsigned long long*)w.scanws;
    const int nz = (int)(scan_status_bytes(m) / sizeof(unsigned long long));
    if (neigh_lean())
        neigh_window<0, true><<<ceil_div(m, 4 * G), 256, neigh_lds_pad(), s>>>(centres, centre_batch_ids, m, sorted_pts, cell_indexs,
                                                     aabb_min, aabb_max, batch_size, num_cells, radius, scale_inv, centre_order,
                                                     w.cnt, w.masks, nullptr, nullptr, 0, zw, nz, G, Tabs, nullptr, nullptr, nullptr, nullptr);
    else
        neigh_window<0, false><<<ceil_div(m, 4 * G), 256, neigh_lds_pad(), s>>>(centres, centre_batch_ids, m, sorted_pts, cell_indexs,
                                                     aabb_min, aabb_max, batch_size, num_cells, radius, scale_inv, centre_order,
                                                     w.cnt, w.masks, nullptr, nullptr, 0, zw, nz, G, Tabs, nullptr, nullptr, nullptr, nullptr);
    MCCNN_LAUNCHED();
    if (skip_scan) return 0;   // (the fill pass that follows in the same chain scans the counts itself: find_neighbors_fill_impl)
    int rc = exclusive_scan_i32(w.cnt, start_idx, m, total_dev, w.scanws, s, true, total_host);
    if (rc) return rc;
    return 0;
}

static int find_neighbors_fill_impl(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                              int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                              int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                              const int* start_idx, int e, int* packed, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream, int* scan_start_out, int* scan_total_dev, int* scan_total_host) {
    if (m < 0 || n < 0 || e < 0 || batch_size <= 0 || num_cells <= 0 || !(radius > 0.0f)) return MCCNN_E_BADARG;
    const bool scan = scan_start_out != nullptr;   // the counts of a skip_scan count pass are still in `ws`: scanned here
    if (scan && (m > MCCNN_NW_SCAN_M || !scan_total_dev)) return MCCNN_E_BADARG;
    if (m == 0 || e == 0) return 0;
    if (!centres || !centre_batch_ids || !sorted_pts || !cell_indexs || !aabb_min || !aabb_max || (!scan && !start_idx) || !packed)
        return MCCNN_E_BADARG;
    NeighWs w;
    if (!neigh_ws(ws, ws_bytes, m, n, w)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    static const int forcedFill = debug_int("nw_group_fill", 0);  // A/B switch
    // (mask rows are indexed by visiting position: the two passes may group differently -- the compaction gains little
    // from more centres per wave: 100 k centres 0.0612 ms at 8, 0.0622 at 16; 800 k centres 0.305 at 8, 0.295 at 24)
    int G = neigh_group(m);
    if (G == 16) G = MCCNN_NW_G;
    if (forcedFill >= 1 && forcedFill <= 32) G = forcedFill;
    const float Tabs = scale_inv ? 0.0f : sqrt_threshold_host(radius);
    size_t dyn = neigh_lds_pad();
    if (scan && dyn < (size_t)m * sizeof(int)) dyn = (size_t)m * sizeof(int);
    const int* scanCnt = scan ? w.cnt : nullptr;
    if (neigh_lean())
        neigh_window<1, true><<<ceil_div(m, 4 * G), 256, dyn, s>>>(centres, centre_batch_ids, m, sorted_pts, cell_indexs,
                                                     aabb_min, aabb_max, batch_size, num_cells, radius, scale_inv, centre_order,
                                                     nullptr, w.masks, start_idx, packed, e, nullptr, 0, G, Tabs, scanCnt, scan_start_out,
                                                     scan_total_dev, scan_total_host);
    else
        neigh_window<1, false><<<ceil_div(m, 4 * G), 256, dyn, s>>>(centres, centre_batch_ids, m, sorted_pts, cell_indexs,
                                                     aabb_min, aabb_max, batch_size, num_cells, radius, scale_inv, centre_order,
                                                     nullptr, w.masks, start_idx, packed, e, nullptr, 0, G, Tabs, scanCnt, scan_start_out,
                                                     scan_total_dev, scan_total_host);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_find_neighbors_fill(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                              int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                              int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                              const int* start_idx, int e, int* packed, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream) {
    return find_neighbors_fill_impl(centres, centre_batch_ids, m, sorted_pts, n, cell_indexs, aabb_min, aabb_max, batch_size,
                                    num_cells, radius, scale_inv, centre_order, start_idx, e, packed, ws, ws_bytes, stream, nullptr,
                                    nullptr, nullptr);
}

}  // extern "C"
namespace mccnn {
// Both passes of a search back to back (the native executor's geometry chain). Lists of at most MCCNN_NW_SCAN_M centres:
// count -> fill, the prefix sum of the counts rides in the fill pass (two launches); larger ones: count -> scan -> fill.
int find_neighbors_chain(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts, int n,
                         const int* cell_indexs, const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                         float radius, int scale_inv, const int* centre_order, int* start_idx, int e_capacity, int* packed,
                         int* total_dev, int* total_host, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    static const bool fusedScan = debug_int("nw_fused", 1) != 0;   // A/B switch, read once
    const bool small = fusedScan && m > 0 && m <= MCCNN_NW_SCAN_M && e_capacity > 0 && n > 0;
    int rc = find_neighbors_count_impl(centres, centre_batch_ids, m, sorted_pts, n, cell_indexs, aabb_min, aabb_max, batch_size,
                                       num_cells, radius, scale_inv, centre_order, start_idx, total_dev, total_host, ws, ws_bytes,
                                       stream, small);
    if (rc) return rc;
    return find_neighbors_fill_impl(centres, centre_batch_ids, m, sorted_pts, n, cell_indexs, aabb_min, aabb_max, batch_size,
                                    num_cells, radius, scale_inv, centre_order, start_idx, e_capacity, packed, ws, ws_bytes, stream,
                                    small ? start_idx : nullptr, small ? total_dev : nullptr, small ? total_host : nullptr);
}
}  // namespace mccnn
extern "C" {

}  // extern "C"
namespace mccnn {
// ---- host side of the batch form (mccnn_geometry_build_batch): one item per search / KDE, the workspace layout of the
// single calls (neigh_ws); background settings (plain loop, centres per wave by the list's size)
bool neigh_batch_eligible(int m, int n) { return m > 0 && n > 0 && (long long)m <= 2048LL * 1024; }
int neigh_batch_item(NeighItem& it, ScanItem& sc, const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                     int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                     float radius, int scale_inv, const int* order, int* start_idx, int e_capacity, int* packed, int* total_dev,
                     int* total_host, void* ws, size_t ws_bytes) {
    if (!neigh_batch_eligible(m, n)) return MCCNN_E_TOOLARGE;
    NeighWs w;
    if (!neigh_ws(ws, ws_bytes, m, n, w)) return MCCNN_E_WORKSPACE;
    const int tiles = ceil_div(m, 2048);
    const int G = m >= 32768 ? MCCNN_NW_G : (m >= 16384 ? 4 : (m >= 8192 ? 2 : 1));   // (neigh_group of background launches)
    it = NeighItem{centres, centre_batch_ids, sorted_pts, cell_indexs, aabb_min, aabb_max, order, w.cnt, w.masks, start_idx, packed,
                   reinterpret_cast<unsigned long long*>(w.scanws), m, batch_size, num_cells, scale_inv, e_capacity, G, tiles + 1,
                   radius, scale_inv ? 0.0f : sqrt_threshold_host(radius)};
    sc = ScanItem{w.cnt, start_idx, reinterpret_cast<unsigned long long*>(w.scanws), total_dev, total_host, m, tiles};
    return 0;
}
int launch_neigh_batch(const NeighBatch& nbt, int count, int mode, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) { bb.first[k] = run; run += ceil_div(nbt.it[k].m, 4 * nbt.it[k].G); }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    if (mode == 0) neigh_window_batch<0><<<run, 256, 24000, s>>>(nbt, bb);   // (the LDS pad of background launches: neigh_lds_pad)
    else neigh_window_batch<1><<<run, 256, 24000, s>>>(nbt, bb);
    MCCNN_LAUNCHED();
    return 0;
}
void pdf_batch_item(PdfItem& it, const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m, const int* packed,
                    int e_capacity, const int* e_dev, const float* aabb_min, const float* aabb_max, int batch_size, float window,
                    float radius, int scale_inv, float* pdfs) {
    it = PdfItem{sorted_pts, sorted_batch_ids, reinterpret_cast<const int2*>(packed), start_idx, aabb_min, aabb_max, pdfs, e_dev,
                 m, e_capacity, batch_size, scale_inv, m >= 16384 ? MCCNN_PDF_ROWS : 1, window, radius};
}
int launch_pdf_batch(const PdfBatch& pb, int count, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) { bb.first[k] = run; run += ceil_div(pb.it[k].m, 16 * pb.it[k].rowsPerWave); }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    pdf_rows_mfma_batch<<<run, 1024, 0, s>>>(pb, bb);
    MCCNN_LAUNCHED();
    return 0;
}
}  // namespace mccnn
extern "C" {

int mccnn_invert_permutation(const int* new_idx, int n, int* inv, mccnn_stream_t stream) {
    if (n < 0) return MCCNN_E_BADARG;
    if (n == 0) return 0;
    if (!new_idx || !inv) return MCCNN_E_BADARG;
    invert_perm_k<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(new_idx, n, inv);
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_compute_pdf_workspace_bytes(int e, int mode) {
    return (mode == 2 && e > 0) ? align_up((size_t)e * sizeof(float4)) : 256;
}

static int compute_pdf_impl(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m,
                            const int* packed, int e, const int* e_dev, const float* aabb_min, const float* aabb_max,
                            int batch_size, float window, float radius, int scale_inv, int mode, float* pdfs, void* ws,
                            size_t ws_bytes, mccnn_stream_t stream) {
    if (m < 0 || e < 0 || batch_size <= 0 || !(radius > 0.0f) || !(window > 0.0f)) return MCCNN_E_BADARG;
    if (e == 0) return 0;
    if (!sorted_pts || !sorted_batch_ids || !start_idx || !packed || !aabb_min || !aabb_max || !pdfs || m == 0)
        return MCCNN_E_BADARG;
    if (e_dev && mode == 0) return MCCNN_E_BADARG;  // the device-count form exists for the single-precision KDE only
    hipStream_t s = (hipStream_t)stream;
    const int2* pk = reinterpret_cast<const int2*>(packed);
    if (mode == 0)
        pdf_edges_ref<<<ceil_div(e, 256), 256, 0, s>>>(sorted_pts, sorted_batch_ids, start_idx, m, pk, e, aabb_min,
                                                      aabb_max, batch_size, window, radius, scale_inv, pdfs);
    else {
        if (mode == 2) {
            if (!ws || ws_bytes < mccnn_compute_pdf_workspace_bytes(e, mode)) return MCCNN_E_WORKSPACE;
            float4* sc = (float4*)ws;
            pdf_edge_coords<<<ceil_div(e, 256), 256, 0, s>>>(sorted_pts, sorted_batch_ids, pk, e, aabb_min, aabb_max,
                                                            batch_size, window, radius, scale_inv, sc, e_dev);
            MCCNN_LAUNCHED();
            pdf_rows<<<ceil_div(m, 4), 256, 0, s>>>(sc, start_idx, m, e, window, pdfs, e_dev);
        } else {
            // rows per wave: 4 consecutive rows let the next row's points fly under this row's tiles, but a list with few
            // (long) rows needs the waves -- 279 centres of 141 neighbours (BASELINE cfg1 Conv_2) ran 123 us on 70 waves
            // (and 16 waves per workgroup: a row longer than the tile planes is walked by all of them)
            if (m >= 16384)
                pdf_rows_mfma<4><<<ceil_div(m, 4 * MCCNN_PDF_ROWS), 256, 0, s>>>(sorted_pts, sorted_batch_ids, pk, start_idx, m, e, aabb_min,
                                                                                aabb_max, batch_size, window, radius, scale_inv, pdfs,
                                                                                e_dev, MCCNN_PDF_ROWS);
            else
                pdf_rows_mfma<16><<<ceil_div(m, 16), 1024, 0, s>>>(sorted_pts, sorted_batch_ids, pk, start_idx, m, e, aabb_min, aabb_max,
                                                                  batch_size, window, radius, scale_inv, pdfs, e_dev, 1);
        }
    }
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_compute_pdf(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m,
                      const int* packed, int e, const float* aabb_min, const float* aabb_max, int batch_size,
                      float window, float radius, int scale_inv, int mode, float* pdfs, void* ws, size_t ws_bytes,
                      mccnn_stream_t stream) {
    return compute_pdf_impl(sorted_pts, sorted_batch_ids, start_idx, m, packed, e, nullptr, aabb_min, aabb_max, batch_size,
                            window, radius, scale_inv, mode, pdfs, ws, ws_bytes, stream);
}

int mccnn_compute_pdf_dn(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m,
                         const int* packed, int e_capacity, const int* e_dev, const float* aabb_min,
                         const float* aabb_max, int batch_size, float window, float radius, int scale_inv, float* pdfs,
                         void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!e_dev) return MCCNN_E_BADARG;
    return compute_pdf_impl(sorted_pts, sorted_batch_ids, start_idx, m, packed, e_capacity, e_dev, aabb_min, aabb_max,
                            batch_size, window, radius, scale_inv, 1, pdfs, ws, ws_bytes, stream);
}

}  // extern "C"
