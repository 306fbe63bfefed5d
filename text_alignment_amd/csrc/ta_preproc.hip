// ta_preproc.hip -- page preprocessing primitives on the GPU: the
// image operations the reference delegates to the Gamera toolkit in textAlignPreprocessing.py
// (to_onebit, despeckle, cc_analysis, rotation_angle_projections, rotate, filter_short_runs /
// filter_narrow_runs, projection_rows; reference :167-195, :212-253), as the host restatement
// text_alignment_amd/textAlignPreprocessing.py expresses them with numpy / scipy.ndimage.  Gamera is
// absent, so this is parity-unpinned against the reference; it is pinned to the host restatement
// (same component sets, same angle, same rotated bits).
//
// One page at a time; images are uint8 planes (ink = 1).  All kernels are plain streaming passes
// (HBM-bound, a few MB per page); connected components use label equivalence: every ink pixel
// starts as its own label (its linear index), a scan pass lowers the root of a pixel's label to the
// smallest label among its 8 neighbours, an analysis pass flattens the label trees, until nothing
// changes (tiles of 16 x 64 pixels are labelled in LDS first, the global passes only stitch across
// tile borders).  The final label of a component is the linear index of its first pixel in raster
// order.
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>

#include "ta_common.h"

namespace ta {

constexpr int kPpThreads = 256;

// Byte planes are walked 16 bytes per lane where the plane starts on a 16-byte boundary (every plane the package
// allocates does; a caller's view may not: then, and for the last n % 16 bytes, one byte per lane as before).
__device__ __forceinline__ bool pp_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// A wave has its own 256 bins (a text page is mostly paper: with one table per workgroup four waves queued on the
// same few bins), merged at the end.
__device__ __forceinline__ void pp_hist_kernel_body(const uint8_t* img, int64_t n, uint32_t* hist) {
    __shared__ uint32_t sh[kPpThreads / 64][256];
    for (int k = threadIdx.x; k < (kPpThreads / 64) * 256; k += kPpThreads) (&sh[0][0])[k] = 0;
    __syncthreads();
    uint32_t* mine = sh[threadIdx.x >> 6];
    const int64_t gid = (int64_t)blockIdx.x * kPpThreads + threadIdx.x, span = (int64_t)gridDim.x * kPpThreads;
    int64_t done = 0;
    if (pp_aligned16(img)) {
        const int64_t n16 = n >> 4;
        const uint4* v = reinterpret_cast<const uint4*>(img);
        for (int64_t i = gid; i < n16; i += span) {
            const uint4 q = v[i];
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&mine[wds[k] & 255u], 1u); atomicAdd(&mine[(wds[k] >> 8) & 255u], 1u);
                atomicAdd(&mine[(wds[k] >> 16) & 255u], 1u); atomicAdd(&mine[wds[k] >> 24], 1u);
            }
        }
        done = n16 << 4;
    }
    for (int64_t e = done + gid; e < n; e += span) atomicAdd(&mine[img[e]], 1u);
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int wv = 0; wv < kPpThreads / 64; ++wv) total += sh[wv][threadIdx.x];
    if (total) atomicAdd(&hist[threadIdx.x], total);
}
static_assert(kPpThreads == 256, "one bin per thread at the end");

// ink = (img <= thr), or its complement
__device__ __forceinline__ void pp_threshold_kernel_body(const uint8_t* img, int64_t n, int thr,
                                                                  int invert, uint8_t* ink) {
    const int64_t gid = (int64_t)blockIdx.x * kPpThreads + threadIdx.x, span = (int64_t)gridDim.x * kPpThreads;
    int64_t done = 0;
    if (pp_aligned16(img) && pp_aligned16(ink)) {
        const int64_t n16 = n >> 4;
        const uint4* v = reinterpret_cast<const uint4*>(img);
        uint4* o = reinterpret_cast<uint4*>(ink);
        for (int64_t i = gid; i < n16; i += span) {
            const uint4 q = v[i];
            uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t r = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int px = (int)((wds[k] >> (8 * b)) & 255u) <= thr;
                    r |= (uint32_t)(invert ? !px : px) << (8 * b);
                }
                wds[k] = r;
            }
            o[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
        done = n16 << 4;
    }
    for (int64_t e = done + gid; e < n; e += span) {
        const int v = img[e] <= thr;
        ink[e] = (uint8_t)(invert ? !v : v);
    }
}

// First stage of the labelling: every 16 x 64 tile is labelled on its own in LDS (the same scan /
// flatten rounds, on tile-local indices, until the tile is stable), so the global rounds only have
// to stitch components across tile borders -- a couple of rounds instead of one per pixel of
// distance.  Output: lab[p] = global index of the tile-local root of p (roots point to themselves).
constexpr int kTileH = 16, kTileW = 64;
__global__ __launch_bounds__(kTileH * kTileW) void pp_label_tile_kernel(const uint8_t* ink, int h, int w, int32_t* lab) {
    __shared__ int loc[kTileH * kTileW];
    const int tx = threadIdx.x % kTileW, ty = threadIdx.x / kTileW;
    const int x = blockIdx.x * kTileW + tx, y = blockIdx.y * kTileH + ty;
    const bool inside = x < w && y < h;
    const int me = threadIdx.x;
    const bool on = inside && ink[(int64_t)y * w + x];
    // a tile row is one wave: every ink pixel starts at the FIRST pixel of its horizontal run (the highest
    // background bit below it in the row's ballot, plus one), so the rounds below only have to join runs
    // of neighbouring rows -- a fully inked tile (the page background once the image is inverted) needs
    // log2(16) rounds instead of one per pixel of width
    static_assert(kTileW == 64, "a tile row is a wave");
    const unsigned long long rowmask = __ballot(on);
    const unsigned long long gaps_below = ~rowmask & ((1ull << tx) - 1ull);
    const int run0 = gaps_below ? 64 - (int)__builtin_clzll(gaps_below) : 0;
    loc[me] = on ? ty * kTileW + run0 : -1;
    __syncthreads();
    // join the runs of neighbouring rows: one lock-free union-find pass over LDS (a root is an entry that points
    // to itself; the smaller index wins, so a component's root ends up its raster-first pixel).  A pixel links to
    // the row above only where no neighbour's link implies it: to the pixel straight above if that is ink (the
    // diagonals are then in the same upper run or background), else to either diagonal that is; and not at all
    // if its left neighbour is ink and sees ink straight above too (same two runs).
    auto find = [&](int a) { while (true) { const int p = loc[a]; if (p == a) return a; a = p; } };
    auto unite = [&](int a, int b) {
        while (true) {
            a = find(a); b = find(b);
            if (a == b) return;
            if (a > b) { const int t = a; a = b; b = t; }
            const int old = atomicMin(&loc[b], a);
            if (old == b) return;
            b = old;
        }
    };
    if (on && ty > 0) {
        const int up = me - kTileW;
        const bool u = loc[up] >= 0;
        const bool ul = tx > 0 && loc[up - 1] >= 0, ur = tx + 1 < kTileW && loc[up + 1] >= 0;
        if (u) {
            const bool left_same = tx > 0 && loc[me - 1] >= 0 && ul;
            if (!left_same) unite(me, up);
        } else {
            if (ul && !(tx > 0 && loc[me - 1] >= 0)) unite(me, up - 1);     // (a left neighbour has it straight above)
            if (ur) unite(me, up + 1);
        }
    }
    __syncthreads();
    if (on) loc[me] = find(me);
    __syncthreads();
    if (inside) {
        int32_t out = -1;
        if (on) {
            const int r = loc[me];
            out = (int32_t)((int64_t)(blockIdx.y * kTileH + r / kTileW) * w + blockIdx.x * kTileW + r % kTileW);
        }
        lab[(int64_t)y * w + x] = out;
    }
}

// Second stage: stitch the tile-local components across tile borders with a lock-free union-find on the
// label array itself (a root is a pixel whose label is its own index; the smaller index always wins, so a
// component's final root is its raster-first pixel whatever the order of the unions).  Only pixels on a
// tile's left / right column or bottom row have neighbours in another tile; each unites itself with its
// "forward" neighbours (right, down-left, down, down-right) that lie in another tile, which covers every
// cross-tile pair once.  ONE pass, no convergence loop and no host round trip: the first form (every
// pixel lowering its root towards its smallest neighbour, then a flatten, repeated until a pass changed
// nothing -- ~4 scan / flatten pairs and 2 host waits per labelling) was half of a page's kernel time.
__device__ __forceinline__ int32_t uf_find(int32_t* lab, int32_t a) {
    // with path halving: a node is re-pointed to its grandparent on the way (atomicMin: parents only ever
    // decrease, so this can only shorten chains) -- the page background, once the image is inverted, is one
    // component across thousands of tiles, and without it every union walks an ever longer chain
    while (true) {
        const int32_t p = __hip_atomic_load(&lab[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == a) return a;
        const int32_t gp = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp == p) return p;
        atomicMin(&lab[a], gp);
        a = gp;
    }
}
__device__ __forceinline__ void uf_unite(int32_t* lab, int32_t a, int32_t b) {
    while (true) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a > b) { const int32_t t = a; a = b; b = t; }          // the larger root goes under the smaller
        const int32_t old = atomicMin(&lab[b], a);
        if (old == b) return;                                       // b was still a root: linked
        b = old;                                                    // b had been linked elsewhere meanwhile: unite with that
    }
}
__device__ __forceinline__ void uf_link(int32_t* lab, int64_t a, int64_t b) {
    // cheap look first, through the (possibly stale) L1: parents only ever move to smaller ancestors, so two
    // walks that meet in one node prove the pixels connected whatever else is going on
    int32_t ra = (int32_t)a, rb = (int32_t)b;
    while (lab[ra] != ra) ra = lab[ra];
    while (lab[rb] != rb) rb = lab[rb];
    if (ra != rb) uf_unite(lab, ra, rb);
}
__global__ __launch_bounds__(kPpThreads) void pp_label_merge_kernel(int32_t* lab, int h, int w) {
    // Work items: the ink pixels of every tile's right column, bottom row and left column.  A link is made
    // only where no other link implies it: the straight neighbour across the border if it is ink (its own
    // in-tile or cross-tile neighbours then carry the diagonals), the diagonals only where the straight one is
    // background; and of a run of border pixels that all face ink only the first links (the others join the
    // same two tile components).
    // Only those pixels are enumerated (a tenth of the page): first the bottom rows of the tile rows, whole; then
    // the left and right columns of the tile columns, without the pixels the bottom rows already had.
    const int ntx = (w + kTileW - 1) / kTileW;
    const int64_t nrow_items = (int64_t)(h / kTileH) * w;                   // rows y = 16 k + 15 < h
    const int64_t ncol_items = (int64_t)h * 2 * ntx;
    const int64_t nitems = nrow_items + ncol_items;
    for (int64_t it = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; it < nitems; it += (int64_t)gridDim.x * kPpThreads) {
        int y, x;
        if (it < nrow_items) {
            y = (int)(it / w) * kTileH + kTileH - 1; x = (int)(it % w);
        } else {
            const int64_t b = it - nrow_items;
            y = (int)(b / (2 * ntx));
            const int t = (int)(b % (2 * ntx));
            x = (t >> 1) * kTileW + ((t & 1) ? kTileW - 1 : 0);
            if (x >= w || (y % kTileH) == kTileH - 1) continue;
        }
        const int64_t e = (int64_t)y * w + x;
        const int cx = x % kTileW, cy = y % kTileH;
        if (lab[e] < 0) continue;
        auto ink = [&](int yy, int xx) { return lab[(int64_t)yy * w + xx] >= 0; };
        if (cx == kTileW - 1 && x + 1 < w) {                           // the tile to the right
            if (ink(y, x + 1)) {
                if (!(cy != 0 && ink(y - 1, x) && ink(y - 1, x + 1))) uf_link(lab, e, e + 1);
            } else if (y + 1 < h && ink(y + 1, x + 1)) {
                uf_link(lab, e, e + w + 1);
            }
        }
        if (cy == kTileH - 1 && y + 1 < h) {                           // the tile below
            if (ink(y + 1, x)) {
                if (!(cx != 0 && ink(y, x - 1) && ink(y + 1, x - 1))) uf_link(lab, e, e + w);
            } else {
                if (x > 0 && ink(y + 1, x - 1)) uf_link(lab, e, e + w - 1);
                if (x + 1 < w && ink(y + 1, x + 1)) uf_link(lab, e, e + w + 1);
            }
        } else if (cx == 0 && x > 0 && y + 1 < h && !ink(y + 1, x) && ink(y + 1, x - 1)) {
            uf_link(lab, e, e + w - 1);                                // down-left into the tile to the left
        }
    }
}

// (and the statistics entries of the roots -- the only ones ever read -- start from their neutral values here:
// filling all five planes, 20 bytes per pixel, was a sixth of a labelling call)
__global__ __launch_bounds__(kPpThreads) void pp_label_flatten_kernel(int32_t* lab, int64_t n, int32_t* stats) {
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) {
        int32_t r = lab[e];
        if (r < 0) continue;
        while (lab[r] != r) r = lab[r];
        lab[e] = r;
        if (r == e && stats) {
            stats[e] = 0;
            stats[n + e] = 0x7fffffff; stats[2 * n + e] = 0x7fffffff;
            stats[3 * n + e] = -1; stats[4 * n + e] = -1;
        }
    }
}

// per-root statistics: area and bounding box (arrays indexed by the root's linear index).  The 64
// pixels of a wave are neighbours in a row and mostly share a root (the page background is one
// component of millions of pixels), so each wave first combines its lanes per distinct root and
// only the first lane of each root goes to memory.
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
// Per-component area and bounding box.  A wave first combines its lanes per distinct root (the page
// background -- one component of millions of pixels once the image is inverted -- would otherwise
// serialise on five atomics); the combined items then go into a small hash table in LDS, one per
// workgroup, and only what a workgroup has gathered over ALL its pixels goes to memory at the end: a
// few atomics per component and workgroup instead of per component and wave-load (the stats kernel was
// a third of a page's device time, 0.28 ms per call, almost all of it contended atomics on the
// background's five words).  A table that fills up (more than kStatSlots distinct roots in one
// workgroup's pixels) sends the overflow straight to memory, as before.
constexpr int kStatSlots = 256;
__global__ __launch_bounds__(kPpThreads) void pp_stats_kernel(const int32_t* lab, int h, int w, int32_t* area,
                                                              int32_t* x0, int32_t* y0, int32_t* x1, int32_t* y1) {
    __shared__ int32_t t_key[kStatSlots], t_area[kStatSlots], t_x0[kStatSlots], t_y0[kStatSlots],
        t_x1[kStatSlots], t_y1[kStatSlots];
    for (int k = threadIdx.x; k < kStatSlots; k += kPpThreads) {
        t_key[k] = -1; t_area[k] = 0; t_x0[k] = 0x7fffffff; t_y0[k] = 0x7fffffff; t_x1[k] = -1; t_y1[k] = -1;
    }
    __syncthreads();
    const int64_t n = (int64_t)h * w;
    const int64_t span = (int64_t)gridDim.x * kPpThreads;
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * kPpThreads; base < n; base += span) {     // uniform trip count per wave
        const int64_t e = base + threadIdx.x;
        int32_t r = (e < n) ? lab[e] : -1;
        unsigned long long todo = __ballot(r >= 0);
        if (todo == 0ull) continue;                           // a wave of background (most of a text page): no coordinates needed
        const uint32_t e32 = (uint32_t)(e < n ? e : n - 1);   // n < 2^31 (checked by the callers)
        const int y = (int)(e32 / (uint32_t)w), x = (int)(e32 - (uint32_t)y * (uint32_t)w);
        // the 64 pixels are neighbours in one row unless the wave wraps a row end
        const bool one_row = __builtin_amdgcn_readlane(y, 0) == __builtin_amdgcn_readlane(y, 63);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int32_t root = __shfl(r, leader, 64);
            const bool same = (r == root);
            const unsigned long long grp = __ballot(same);
            int mnx, mny, mxx, mxy;
            if (one_row) {                                    // x grows with the lane: the group's ends are its first and last lane
                mnx = __shfl(x, leader, 64);
                mxx = __shfl(x, 63 - (int)__builtin_clzll(grp), 64);
                mny = mxy = __builtin_amdgcn_readlane(y, 0);
            } else {
                mnx = wave_min(same ? x : 0x7fffffff); mny = wave_min(same ? y : 0x7fffffff);
                mxx = wave_max(same ? x : -1); mxy = wave_max(same ? y : -1);
            }
            if (lane == leader) {
                const int cnt = (int)__popcll(grp);
                unsigned slot = ((unsigned)root * 2654435761u) >> 24;            // 8 bits: kStatSlots = 256
                int found = -1;
                for (int probe = 0; probe < 8; ++probe) {
                    const int32_t was = atomicCAS(&t_key[slot], -1, root);
                    if (was == -1 || was == root) { found = (int)slot; break; }
                    slot = (slot + 1) & (kStatSlots - 1);
                }
                if (found >= 0) {
                    atomicAdd(&t_area[found], cnt);
                    atomicMin(&t_x0[found], mnx); atomicMin(&t_y0[found], mny);
                    atomicMax(&t_x1[found], mxx); atomicMax(&t_y1[found], mxy);
                } else {
                    atomicAdd(&area[root], cnt);
                    atomicMin(&x0[root], mnx); atomicMin(&y0[root], mny);
                    atomicMax(&x1[root], mxx); atomicMax(&y1[root], mxy);
                }
            }
            todo &= ~grp;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kStatSlots; k += kPpThreads) {
        const int32_t root = t_key[k];
        if (root < 0) continue;
        atomicAdd(&area[root], t_area[k]);
        atomicMin(&x0[root], t_x0[k]); atomicMin(&y0[root], t_y0[k]);
        atomicMax(&x1[root], t_x1[k]); atomicMax(&y1[root], t_y1[k]);
    }
}
static_assert(kStatSlots == 256, "the hash keeps 8 bits");
constexpr int kStatBlocks = 1024;         // workgroups of the stats kernel (grid-stride over the page)

// records {root, area, x0, y0, x1, y1} of every component, in no particular order
__global__ __launch_bounds__(kPpThreads) void pp_collect_kernel(const int32_t* lab, int64_t n, const int32_t* area,
                                                                const int32_t* x0, const int32_t* y0,
                                                                const int32_t* x1, const int32_t* y1,
                                                                int32_t* recs, int32_t cap, int32_t* count) {
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) {
        if (lab[e] != (int32_t)e) continue;
        const int k = atomicAdd(count, 1);
        if (k < cap) {
            int32_t* r = recs + (int64_t)k * 6;
            r[0] = (int32_t)e; r[1] = area[e]; r[2] = x0[e]; r[3] = y0[e]; r[4] = x1[e]; r[5] = y1[e];
        }
    }
}

// ink &= keep(component): area >= min_area and height <= max_height (either bound may be off)
__global__ __launch_bounds__(kPpThreads) void pp_filter_kernel(uint8_t* ink, const int32_t* lab, int64_t n,
                                                               const int32_t* area, const int32_t* y0,
                                                               const int32_t* y1, int min_area, int max_height) {
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) {
        const int32_t r = lab[e];
        if (r < 0) continue;
        const bool keep = area[r] >= min_area && (y1[r] - y0[r] + 1) <= max_height;
        if (!keep) ink[e] = 0;
    }
}

__global__ __launch_bounds__(kPpThreads) void pp_invert_kernel(uint8_t* ink, int64_t n) {
    const int64_t gid = (int64_t)blockIdx.x * kPpThreads + threadIdx.x, span = (int64_t)gridDim.x * kPpThreads;
    int64_t done = 0;
    if (pp_aligned16(ink)) {
        const int64_t n16 = n >> 4;
        uint4* v = reinterpret_cast<uint4*>(ink);
        // per byte !b: 0x01 where the byte is zero, 0x00 where it is not (whatever non-zero value it holds)
        auto inv = [](uint32_t x) -> uint32_t {
            const uint32_t nz = ((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;
            return nz ^ 0x01010101u;
        };
        for (int64_t i = gid; i < n16; i += span) {
            const uint4 q = v[i];
            v[i] = make_uint4(inv(q.x), inv(q.y), inv(q.z), inv(q.w));
        }
        done = n16 << 4;
    }
    for (int64_t e = done + gid; e < n; e += span) ink[e] = !ink[e];
}

// row histogram of the page rotated by each candidate angle, from the ink coordinates of the
// decimated page: pixel (y, x) lands on row rint(cy + dy cos a - dx sin a).
// Workgroup (c, a) histograms slice c of the pixels for angle a in LDS and adds its non-empty bins to
// hist[a][*] at the end (the first form -- one global atomic per ink pixel and angle, 3 M per sweep --
// took 0.3 ms per sweep, a fifth of a page's device time).  More rows than the LDS histogram holds:
// straight to memory, as before.
constexpr int kAngleBins = 4096;
constexpr int kAngleRun = 16;             // consecutive decimated pixels of one row per thread
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// A thread takes kAngleRun consecutive pixels of one decimated row: at these angles (|a| <= 6 degrees)
// they land on one or two rows, so it counts runs of equal rows and the lanes of a wave -- neighbouring
// runs, mostly on the same rows again -- combine equal rows before anything touches the histogram: one
// LDS atomic per distinct row and wave instead of one per ink pixel (same-address LDS atomics of a wave
// serialise, and a text line puts most of a wave's ink on one bin).
__global__ __launch_bounds__(kPpThreads) void pp_angle_hist_kernel(const uint8_t* ink, int h, int w, int step,
                                                                   const double* cs, int nang, uint32_t* hist) {
    __shared__ uint32_t bins[kAngleBins];
    const int hs = (h + step - 1) / step, wsm = (w + step - 1) / step;
    const double cy = (hs - 1) / 2.0, cx = (wsm - 1) / 2.0;
    const int runs_per_row = (wsm + kAngleRun - 1) / kAngleRun;
    const int64_t nitems = (int64_t)hs * runs_per_row;
    const int a = blockIdx.y;
    const bool in_lds = hs <= kAngleBins;
    if (in_lds) {
        for (int k = threadIdx.x; k < hs; k += kPpThreads) bins[k] = 0u;
        __syncthreads();
    }
    const double ca = cs[2 * a], sa = cs[2 * a + 1];
    uint32_t* const out = hist + (int64_t)a * hs;
    const int lane = threadIdx.x & 63;
    const int64_t span = (int64_t)gridDim.x * kPpThreads;
    for (int64_t base = (int64_t)blockIdx.x * kPpThreads; base < nitems; base += span) {   // uniform trip count per wave
        const int64_t item = base + threadIdx.x;
        // up to two (row, count) runs per thread; a third distinct row inside 16 pixels cannot occur for
        // |sin a| * 16 < 2, and is flushed on its own if it ever does
        int row_a = -1, cnt_a = 0, row_b = -1, cnt_b = 0;
        if (item < nitems) {
            const int ys = (int)(item / runs_per_row), x_lo = (int)(item % runs_per_row) * kAngleRun;
            const double dy = __dadd_rn((double)ys, -cy);
            const double t0 = __dadd_rn(cy, __dmul_rn(dy, ca));
            const uint8_t* src = ink + (int64_t)ys * step * w;
            const int x_hi = min(x_lo + kAngleRun, wsm);
            for (int xs = x_lo; xs < x_hi; ++xs) {
                if (!src[(int64_t)xs * step]) continue;
                const double dx = __dadd_rn((double)xs, -cx);
                const double v = __dadd_rn(t0, -__dmul_rn(dx, sa));
                const long long rl = (long long)rint(v);
                if (rl < 0 || rl >= hs) continue;
                const int row = (int)rl;
                if (row == row_a) ++cnt_a;
                else if (row == row_b) ++cnt_b;
                else if (row_a < 0) { row_a = row; cnt_a = 1; }
                else if (row_b < 0) { row_b = row; cnt_b = 1; }
                else { if (in_lds) atomicAdd(&bins[row], 1u); else atomicAdd(&out[row], 1u); }
            }
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int row = which ? row_b : row_a, cnt = which ? cnt_b : cnt_a;
            unsigned long long todo = __ballot(row >= 0);
            while (todo) {
                const int leader = __builtin_ctzll(todo);
                const int r = __shfl(row, leader, 64);
                const bool same = (row == r);
                const unsigned long long grp = __ballot(same);
                const int total = wave_sum(same ? cnt : 0);
                if (lane == leader) { if (in_lds) atomicAdd(&bins[r], (unsigned)total); else atomicAdd(&out[r], (unsigned)total); }
                todo &= ~grp;
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < hs; k += kPpThreads)
            if (bins[k]) atomicAdd(&out[k], bins[k]);
    }
}

// The skew search in two steps: the ink pixels of the decimated page are listed ONCE (a text page is ~8 % ink),
// and every angle of both sweeps then walks the list instead of the page -- 49 + 21 passes over ~55 k points
// instead of over 685 k pixels.  points[i] = (row << 16) | column of the decimated grid, in no particular
// order; *count (device) receives their number.
__device__ __forceinline__ void pp_ink_points_kernel_body(const uint8_t* ink, int h, int w, int step,
                                                                   uint32_t* points, uint32_t* count) {
    const int hs = (h + step - 1) / step, wsm = (w + step - 1) / step;
    const int64_t n = (int64_t)hs * wsm;
    // one append per WORKGROUP and pass (a wave's ballot gives its count, the waves' counts meet in LDS): one
    // atomic on the single counter per 64 pixels was most of this kernel's time
    __shared__ uint32_t wcnt[kPpThreads / 64];
    __shared__ uint32_t wbase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t span = (int64_t)gridDim.x * kPpThreads;
    for (int64_t base = (int64_t)blockIdx.x * kPpThreads; base < n; base += span) {     // uniform trip count per workgroup
        const int64_t e = base + threadIdx.x;
        int ys = 0, xs = 0;
        bool on = false;
        if (e < n) {
            ys = (int)(e / wsm); xs = (int)(e % wsm);
            on = ink[(int64_t)ys * step * w + (int64_t)xs * step] != 0;
        }
        const unsigned long long m = __ballot(on);
        if (lane == 0) wcnt[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int k = 0; k < kPpThreads / 64; ++k) tot += wcnt[k];
            wbase = tot ? atomicAdd(count, tot) : 0u;
        }
        __syncthreads();
        uint32_t at = wbase;
        for (int k = 0; k < wave; ++k) at += wcnt[k];
        if (on) points[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = ((uint32_t)ys << 16) | (uint32_t)xs;
        __syncthreads();                                   // wcnt / wbase are rewritten by the next pass
    }
}

// hist[a][row] from the point list: the arithmetic of pp_angle_hist_kernel per point (float64, same operation order)
__device__ __forceinline__ void pp_angle_hist_points_kernel_body(const uint32_t* points, const uint32_t* count,
                                                                          int hs, int wsm, const double* cs,
                                                                          uint32_t* hist) {
    __shared__ uint32_t bins[kAngleBins];
    const double cy = (hs - 1) / 2.0, cx = (wsm - 1) / 2.0;
    const int a = blockIdx.y;
    const bool in_lds = hs <= kAngleBins;
    if (in_lds) {
        for (int k = threadIdx.x; k < hs; k += kPpThreads) bins[k] = 0u;
        __syncthreads();
    }
    const double ca = cs[2 * a], sa = cs[2 * a + 1];
    uint32_t* const out = hist + (int64_t)a * hs;
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * kPpThreads + threadIdx.x; i < n; i += gridDim.x * kPpThreads) {
        const uint32_t p = points[i];
        const int ys = (int)(p >> 16), xs = (int)(p & 0xFFFFu);
        const double dy = __dadd_rn((double)ys, -cy);
        const double t0 = __dadd_rn(cy, __dmul_rn(dy, ca));
        const double dx = __dadd_rn((double)xs, -cx);
        const double v = __dadd_rn(t0, -__dmul_rn(dx, sa));
        const long long rl = (long long)rint(v);
        if (rl < 0 || rl >= hs) continue;
        if (in_lds) atomicAdd(&bins[(int)rl], 1u); else atomicAdd(&out[(int)rl], 1u);
    }
    if (in_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < hs; k += kPpThreads)
            if (bins[k]) atomicAdd(&out[k], bins[k]);
    }
}

// scipy.ndimage.affine_transform(float32(ink), M, offset, order = 1, mode = 'constant', cval = 0) > 0.5
__device__ __forceinline__ void pp_rotate_kernel_body(const uint8_t* ink, int h, int w, uint8_t* out,
                                                               int oh, int ow, const double* mo) {
    const double m00 = mo[0], m01 = mo[1], m10 = mo[2], m11 = mo[3], off0 = mo[4], off1 = mo[5];
    const int64_t n = (int64_t)oh * ow;
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) {
        const int i = (int)(e / ow), j = (int)(e % ow);
        const double cy = __dadd_rn(__dadd_rn(off0, __dmul_rn((double)i, m00)), __dmul_rn((double)j, m01));
        const double cx = __dadd_rn(__dadd_rn(off1, __dmul_rn((double)i, m10)), __dmul_rn((double)j, m11));
        uint8_t bit = 0;
        if (!(cy < 0.0 || cy > (double)(h - 1) || cx < 0.0 || cx > (double)(w - 1))) {
            const int y0 = (int)floor(cy), x0 = (int)floor(cx);
            const double ty = __dadd_rn(cy, -(double)y0), tx = __dadd_rn(cx, -(double)x0);
            const double wy[2] = {__dadd_rn(1.0, -ty), ty}, wx[2] = {__dadd_rn(1.0, -tx), tx};
            double t = 0.0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = min(y0 + dy, h - 1), xx = min(x0 + dx, w - 1);
                    const double v = ink[(int64_t)yy * w + xx] ? 1.0 : 0.0;
                    t = __dadd_rn(t, __dmul_rn(__dmul_rn(v, wy[dy]), wx[dx]));
                }
            bit = (float)t > 0.5f;
        }
        out[e] = bit;
    }
}

// opening with a line of `len` pixels along the axis: a pixel survives iff some window of `len`
// consecutive pixels containing it is all ink
__device__ __forceinline__ void pp_open_runs_kernel_body(const uint8_t* in, uint8_t* out, int h, int w,
                                                                  int len, int axis) {
    const int64_t n = (int64_t)h * w;
    const int L = axis == 0 ? h : w;
    const int64_t stride = axis == 0 ? w : 1;
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) {
        uint8_t keep = 0;
        if (in[e]) {
            const int pos = axis == 0 ? (int)(e / w) : (int)(e % w);
            int before = 0, after = 0;                      // ink run lengths on either side, capped at len - 1
            while (before < len - 1 && pos - before - 1 >= 0 && in[e - (int64_t)(before + 1) * stride]) ++before;
            while (after < len - 1 && pos + after + 1 < L && in[e + (int64_t)(after + 1) * stride]) ++after;
            keep = (before + after + 1) >= len;
        }
        out[e] = keep;
    }
}

__device__ __forceinline__ void pp_row_sums_kernel_body(const uint8_t* ink, int h, int w, int32_t* sums) {
    __shared__ int sh[kPpThreads];
    const int y = blockIdx.x;
    int acc = 0;
    for (int x = threadIdx.x; x < w; x += kPpThreads) acc += ink[(int64_t)y * w + x];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kPpThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[y] = sh[0];
}

// the text-line strips of a page, cut out of its ink plane into one packed buffer as the greyscale images the
// reference saves for the recogniser (ink black on white): boxes[s] = {ulx, uly, lrx, lry, offset into out}
__device__ __forceinline__ void pp_cut_strips_kernel_body(const uint8_t* __restrict__ ink, int w,
                                                                   const int64_t* __restrict__ boxes,
                                                                   uint8_t* __restrict__ out) {
    const int64_t* b = boxes + 5 * (int64_t)blockIdx.y;
    const int ulx = (int)b[0], uly = (int)b[1], sw = (int)(b[2] - b[0]) + 1, sh = (int)(b[3] - b[1]) + 1;
    uint8_t* o = out + b[4];
    for (int y = blockIdx.x; y < sh; y += gridDim.x) {
        const uint8_t* src = ink + (int64_t)(uly + y) * w + ulx;
        for (int x = threadIdx.x; x < sw; x += kPpThreads) o[(int64_t)y * sw + x] = src[x] ? 0 : 255;
    }
}

__device__ __forceinline__ void pp_clear_rows_kernel_body(uint8_t* ink, int w, const int32_t* rows, int nrows) {
    const int r = rows[blockIdx.x];
    for (int x = threadIdx.x; x < w; x += kPpThreads) ink[(int64_t)r * w + x] = 0;
}

__global__ __launch_bounds__(kPpThreads) void pp_fill_kernel(int32_t* a, int64_t n, int32_t v) {
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) a[e] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Connected components over RUNS.  A text page is nine tenths paper: its ~6 M pixels are a few hundred thousand horizontal
// runs of ink (or, for the hole filling, of paper), and everything a labelling is used for here -- drop the components
// under an area, fill the holes under an area, drop the components over a height, list the components with their boxes --
// needs the components of the runs, not a label per pixel.  One pass finds the runs (a wave per row: the starts and ends
// of a 64-pixel segment's runs are two bit tricks on its ballot), one lock-free union-find pass joins every run with the
// runs of the row above that touch it (8-connectivity: column ranges that overlap after widening one by a pixel), one
// pass flattens and gathers area and bounding box per root (combined per wave and per workgroup first: the page
// background is one component of most of the paper runs).  Runs are numbered in raster order (a scan over the rows'
// counts), the smaller index wins a union, so a component's root is its raster-first run and the root's first pixel the
// component's raster-first pixel -- the label the per-pixel labelling (ta_pp_label) gives it.
struct PpRuns {
    int32_t* row_cnt;    // [h]      runs in each row
    int32_t* row_off;    // [h + 1]  first run of each row; row_off[h] = number of runs
    int32_t* x0;         // [cap]    first / last column of a run, its row, its parent (union-find), and per ROOT:
    int32_t* x1;
    int32_t* yrow;
    int32_t* parent;
    int32_t* area;       //          pixels, and the bounding box
    int32_t* bx0; int32_t* by0; int32_t* bx1; int32_t* by1;
};

// the pages of one stage call, for kernels launched ONCE per batch (blockIdx.y = page): in a stream a page's forty small
// kernels run one after the other whatever the device could hold, so a batch's chain is as long as its launches are many
constexpr int kRunPages = 8;
struct PpRunsBatch {
    const uint8_t* ink[kRunPages];
    int h[kRunPages], w[kRunPages];
    PpRuns R[kRunPages];
};

__device__ __forceinline__ bool pp_run_pixel(const uint8_t* p, int x, int w, int want) { return x < w && ((p[x] != 0) == (want != 0)); }

// (a lane's loads are single bytes, 64 B per instruction and wave: kRunUnroll segments' loads are issued before the first
// ballot, so a row costs a few memory latencies instead of one per 64 pixels)
constexpr int kRunUnroll = 8;
__global__ __launch_bounds__(kPpThreads) void pp_runs_count_kernel(PpRunsBatch B, int want) {
    const uint8_t* ink = B.ink[blockIdx.y]; const int h = B.h[blockIdx.y], w = B.w[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    const int lane = threadIdx.x & 63, row = blockIdx.x * (kPpThreads / 64) + (threadIdx.x >> 6);
    if (row >= h) return;                                    // (a whole wave: the row is the wave's)
    const uint8_t* p = ink + (int64_t)row * w;
    int cnt = 0;
    unsigned long long carry = 0;
    for (int xs = 0; xs < w; xs += 64 * kRunUnroll) {
        bool on[kRunUnroll];
#pragma unroll
        for (int u = 0; u < kRunUnroll; ++u) on[u] = pp_run_pixel(p, xs + 64 * u + lane, w, want);
#pragma unroll
        for (int u = 0; u < kRunUnroll; ++u) {
            const unsigned long long m = __ballot(on[u]);
            cnt += (int)__popcll(m & ~((m << 1) | carry));
            carry = m >> 63;
        }
    }
    if (lane == 0) R.row_cnt[row] = cnt;
}

// exclusive scan of the rows' run counts (one workgroup: a page has a few thousand rows)
__global__ __launch_bounds__(1024) void pp_runs_scan_kernel(PpRunsBatch B) {
    const int h = B.h[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    __shared__ int part[1024];
    const int per = (h + 1023) / 1024, a = threadIdx.x * per, b = min(a + per, h);
    int sum = 0;
    for (int r = a; r < b; ++r) sum += R.row_cnt[r];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                     // Hillis-Steele, inclusive
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int off = part[threadIdx.x] - sum;
    for (int r = a; r < b; ++r) { R.row_off[r] = off; off += R.row_cnt[r]; }
    if (threadIdx.x == 1023) R.row_off[h] = part[1023];
}

__global__ __launch_bounds__(kPpThreads) void pp_runs_write_kernel(PpRunsBatch B, int want) {
    const uint8_t* ink = B.ink[blockIdx.y]; const int h = B.h[blockIdx.y], w = B.w[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    const int lane = threadIdx.x & 63, row = blockIdx.x * (kPpThreads / 64) + (threadIdx.x >> 6);
    if (row >= h) return;
    const uint8_t* p = ink + (int64_t)row * w;
    const int base = R.row_off[row];
    const unsigned long long below = (1ull << lane) - 1ull;
    int nstart = 0, nend = 0;
    unsigned long long carry = 0;
    for (int xs = 0; xs < w; xs += 64 * kRunUnroll) {
        bool on[kRunUnroll + 1];                             // (one more: the ends of a segment's runs need the next one's first pixel)
#pragma unroll
        for (int u = 0; u <= kRunUnroll; ++u) on[u] = pp_run_pixel(p, xs + 64 * u + lane, w, want);
        unsigned long long m = __ballot(on[0]);
#pragma unroll
        for (int u = 0; u < kRunUnroll; ++u) {
            const unsigned long long next = __ballot(on[u + 1]);
            const int x = xs + 64 * u + lane;
            const unsigned long long starts = m & ~((m << 1) | carry);
            if ((starts >> lane) & 1ull) {
                const int i = base + nstart + (int)__popcll(starts & below);
                R.x0[i] = x; R.yrow[i] = row; R.parent[i] = i;
                R.area[i] = 0; R.bx0[i] = 0x7fffffff; R.by0[i] = 0x7fffffff; R.bx1[i] = -1; R.by1[i] = -1;
            }
            nstart += (int)__popcll(starts);
            const unsigned long long ends = m & ~((m >> 1) | ((next & 1ull) << 63));
            if ((ends >> lane) & 1ull) R.x1[base + nend + (int)__popcll(ends & below)] = x;
            nend += (int)__popcll(ends);
            carry = m >> 63;
            m = next;
        }
    }
}

// every run with the runs of the row above whose columns reach its own widened by one: 8-connectivity.  In two passes:
// first inside BANDS of kRunBand rows (no tree grows deeper than a band), a flatten, then across the band borders --
// joined all at once, the runs of the paper (one component over the whole page) hung in chains as long as the page is
// tall, and every later find walked them (42 us a call; the smaller index wins, so every chain leads to the top).
constexpr int kRunBand = 32;
__device__ __forceinline__ void pp_run_join_up(const PpRuns& R, int i, int y) {
    const int lo = R.x0[i] - 1, hi = R.x1[i] + 1;
    int a = R.row_off[y - 1], b = R.row_off[y];
    const int end = b;
    while (a < b) {                                          // first run of the row above that ends at lo or beyond
        const int mid = (a + b) >> 1;
        if (R.x1[mid] < lo) a = mid + 1; else b = mid;
    }
    for (int j = a; j < end && R.x0[j] <= hi; ++j) uf_link(R.parent, i, j);
}
// inside the bands: a workgroup per band, the band's runs and their union-find in LDS (a find is a chain of dependent
// loads, up to a band deep where a stroke -- or the paper -- runs down the page: from LDS a hop is a tenth of what it is
// from L2); a band with more runs than fit takes the same steps on the global arrays.  Leaves parent[] FLAT inside
// every band (each run points at its band's root run).
constexpr int kBandRuns = 4096;           // runs of a band held in LDS: 3 x 4 bytes each (48 KB)
template <bool LDS>
__device__ __forceinline__ void pp_runs_band(const PpRuns& R, int first, int n, int y0, int y1, int* lx0, int* lx1, int* lpar,
                                             const int* roff) {
    // (LDS: lx0 / lx1 / lpar hold the band's runs, indices are band-local; else they are R.x0 / R.x1 / R.parent + first)
    if (LDS) {
        for (int k = threadIdx.x; k < n; k += kPpThreads) { lx0[k] = R.x0[first + k]; lx1[k] = R.x1[first + k]; lpar[k] = k; }
        __syncthreads();
    }
    auto find = [&](int a) { while (true) { const int p_ = lpar[a]; if (p_ == a) return a; a = p_; } };
    for (int i = threadIdx.x; i < n; i += kPpThreads) {       // a thread per run; its row from the band's row offsets (LDS)
        int ya = 0, yb = y1 - y0;                             // largest r with roff[r] <= i
        while (yb - ya > 1) { const int mid = (ya + yb) >> 1; if (roff[mid] <= i) ya = mid; else yb = mid; }
        if (ya == 0) continue;                                // the band's first row: joined across the border later
        const int ra = roff[ya], pa = roff[ya - 1];
        const int lo = lx0[i] - 1, hi = lx1[i] + 1;
        int a = pa, b = ra;
        while (a < b) { const int mid = (a + b) >> 1; if (lx1[mid] < lo) a = mid + 1; else b = mid; }
        for (int j = a; j < ra && lx0[j] <= hi; ++j) {
            int u = i, v = j;
            while (true) {                                    // lock-free union: the smaller index wins
                u = find(u); v = find(v);
                if (u == v) break;
                if (u > v) { const int t = u; u = v; v = t; }
                const int old = atomicMin(&lpar[v], u);
                if (old == v) break;
                v = old;
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += kPpThreads) {
        const int r = find(k);
        if (LDS) R.parent[first + k] = first + r;
        else lpar[k] = r;                                     // (global: parents are band-relative here, made absolute below)
    }
}
__global__ __launch_bounds__(kPpThreads) void pp_runs_union_kernel(PpRunsBatch B) {
    const int h = B.h[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    __shared__ int lx0[kBandRuns], lx1[kBandRuns], lpar[kBandRuns], roff[kRunBand + 1];
    const int y0 = blockIdx.x * kRunBand, y1 = min(y0 + kRunBand, h);
    if (y0 >= h) return;
    const int first = R.row_off[y0], n = R.row_off[y1] - first;
    if (threadIdx.x <= y1 - y0) roff[threadIdx.x] = R.row_off[y0 + threadIdx.x] - first;       // band-relative row offsets
    __syncthreads();
    if (n <= kBandRuns) {
        pp_runs_band<true>(R, first, n, y0, y1, lx0, lx1, lpar, roff);
    } else {
        // too many runs for LDS (a page of noise): the same on the global arrays, with band-relative parents meanwhile
        int* gpar = R.parent + first;
        for (int k = threadIdx.x; k < n; k += kPpThreads) gpar[k] = k;
        __syncthreads();
        pp_runs_band<false>(R, first, n, y0, y1, R.x0 + first, R.x1 + first, gpar, roff);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kPpThreads) gpar[k] += first;
    }
}
__global__ __launch_bounds__(kPpThreads) void pp_runs_union_borders_kernel(PpRunsBatch B) {      // a workgroup per band border
    const int h = B.h[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    const int y = (blockIdx.x + 1) * kRunBand;
    if (y >= h) return;
    for (int i = R.row_off[y] + threadIdx.x; i < R.row_off[y + 1]; i += kPpThreads) pp_run_join_up(R, i, y);
}

// parent[i] = root; area and bounding box of every root (pp_stats_kernel's two-level combination: lanes of a wave
// that share a root first, then a hash table per workgroup, then memory)
__global__ __launch_bounds__(kPpThreads) void pp_runs_stats_kernel(PpRunsBatch B) {
    const int h = B.h[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    __shared__ int32_t t_key[kStatSlots], t_area[kStatSlots], t_x0[kStatSlots], t_y0[kStatSlots],
        t_x1[kStatSlots], t_y1[kStatSlots];
    for (int k = threadIdx.x; k < kStatSlots; k += kPpThreads) {
        t_key[k] = -1; t_area[k] = 0; t_x0[k] = 0x7fffffff; t_y0[k] = 0x7fffffff; t_x1[k] = -1; t_y1[k] = -1;
    }
    __syncthreads();
    const int total = R.row_off[h];
    const int span = gridDim.x * kPpThreads, lane = threadIdx.x & 63;
    auto put = [&](int32_t root, int cnt, int mnx, int mny, int mxx, int mxy) {
        unsigned slot = ((unsigned)root * 2654435761u) >> 24;
        int found = -1;
        for (int probe = 0; probe < 8; ++probe) {
            const int32_t was = atomicCAS(&t_key[slot], -1, root);
            if (was == -1 || was == root) { found = (int)slot; break; }
            slot = (slot + 1) & (kStatSlots - 1);
        }
        if (found >= 0) {
            atomicAdd(&t_area[found], cnt);
            atomicMin(&t_x0[found], mnx); atomicMin(&t_y0[found], mny);
            atomicMax(&t_x1[found], mxx); atomicMax(&t_y1[found], mxy);
        } else {
            atomicAdd(&R.area[root], cnt);
            atomicMin(&R.bx0[root], mnx); atomicMin(&R.by0[root], mny);
            atomicMax(&R.bx1[root], mxx); atomicMax(&R.by1[root], mxy);
        }
    };
    for (int base = blockIdx.x * kPpThreads; base < total; base += span) {          // uniform trip count per wave
        const int i = base + threadIdx.x;
        int32_t r = -1;
        int len = 0, ax0 = 0, ax1 = 0, ay = 0;
        if (i < total) {
            r = i;
            while (R.parent[r] != r) r = R.parent[r];        // (no union is running any more: plain reads)
            R.parent[i] = r;
            ax0 = R.x0[i]; ax1 = R.x1[i]; ay = R.yrow[i]; len = ax1 - ax0 + 1;
        }
        // The runs of a wave are neighbours in raster order.  Paper runs mostly share ONE root (the page background):
        // the wave combines the lanes of its two most frequent-looking roots (the first lane's, then the first other's)
        // before the table; ink runs are letters, a root each: those lanes go to the table on their own, side by side.
        unsigned long long todo = __ballot(r >= 0);
        for (int pass = 0; pass < 2 && todo; ++pass) {
            const int leader = __builtin_ctzll(todo);
            const int32_t root = __shfl(r, leader, 64);
            const bool same = (r == root);
            const unsigned long long grp = __ballot(same);
            if (__popcll(grp) >= 4) {
                int cnt = same ? len : 0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
                const int mnx = wave_min(same ? ax0 : 0x7fffffff), mny = wave_min(same ? ay : 0x7fffffff);
                const int mxx = wave_max(same ? ax1 : -1), mxy = wave_max(same ? ay : -1);
                if (lane == leader) put(root, cnt, mnx, mny, mxx, mxy);
                if (same) r = -1;                             // done
            }
            todo &= ~grp;
        }
        if (r >= 0) put(r, len, ax0, ay, ax1, ay);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kStatSlots; k += kPpThreads) {
        const int32_t root = t_key[k];
        if (root < 0) continue;
        atomicAdd(&R.area[root], t_area[k]);
        atomicMin(&R.bx0[root], t_x0[k]); atomicMin(&R.by0[root], t_y0[k]);
        atomicMax(&R.bx1[root], t_x1[k]); atomicMax(&R.by1[root], t_y1[k]);
    }
}

// the runs of components under min_area pixels or over max_height rows take the value `fill` (0: ink runs dropped;
// 1: paper runs -- holes -- filled)
__global__ __launch_bounds__(kPpThreads) void pp_runs_filter_kernel(PpRunsBatch B, int min_area, int max_height, int fill) {
    uint8_t* ink = const_cast<uint8_t*>(B.ink[blockIdx.y]); const int h = B.h[blockIdx.y], w = B.w[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    const int total = R.row_off[h];
    for (int i = blockIdx.x * kPpThreads + threadIdx.x; i < total; i += gridDim.x * kPpThreads) {
        const int r = R.parent[i];
        const bool keep = R.area[r] >= min_area && (R.by1[r] - R.by0[r] + 1) <= max_height;
        if (keep) continue;
        uint8_t* p = ink + (int64_t)R.yrow[i] * w;
        for (int x = R.x0[i]; x <= R.x1[i]; ++x) p[x] = (uint8_t)fill;
    }
}

// records {root pixel, area, x0, y0, x1, y1} of every component, as pp_collect_kernel writes them
__global__ __launch_bounds__(kPpThreads) void pp_runs_collect_kernel(PpRunsBatch B, int32_t* recs_all, int32_t cap, int32_t* counts) {
    const int h = B.h[blockIdx.y], w = B.w[blockIdx.y]; const PpRuns& R = B.R[blockIdx.y];
    int32_t* recs = recs_all + (size_t)blockIdx.y * cap * 6; int32_t* count = counts + blockIdx.y;
    const int total = R.row_off[h];
    for (int i = blockIdx.x * kPpThreads + threadIdx.x; i < total; i += gridDim.x * kPpThreads) {
        if (R.parent[i] != i) continue;
        const int k = atomicAdd(count, 1);
        if (k < cap) {
            int32_t* r = recs + (int64_t)k * 6;
            r[0] = (int32_t)((int64_t)R.yrow[i] * w + R.x0[i]); r[1] = R.area[i];
            r[2] = R.bx0[i]; r[3] = R.by0[i]; r[4] = R.bx1[i]; r[5] = R.by1[i];
        }
    }
}

// the run tables of a page inside the scratch the labelling entry points are given (lab: h*w int32, stats: 5*h*w):
// a row has at most ceil(w / 2) runs; pages under 8 columns do not fit and take the per-pixel labelling
static bool pp_runs_fit(int h, int w) { return w >= 8 && h >= 1; }
static PpRuns pp_runs_in(int32_t* lab, int32_t* stats, int h, int w) {
    const int64_t cap = (int64_t)h * ((w + 1) / 2);
    PpRuns R;
    R.x0 = stats; R.x1 = stats + cap; R.yrow = stats + 2 * cap; R.parent = stats + 3 * cap; R.area = stats + 4 * cap;
    R.by0 = stats + 5 * cap; R.by1 = stats + 6 * cap; R.bx0 = stats + 7 * cap;
    R.bx1 = lab; R.row_cnt = lab + cap; R.row_off = lab + cap + h;
    return R;
}
constexpr int kRunBlocks = 256;           // workgroups per page of the per-run kernels (grid-stride; the run count lives on the device)
// runs of value `want` of the batch's pages, joined into components with their statistics: six launches for ALL pages of
// the batch (blockIdx.y = page; grids sized for the tallest), nothing waited for
static void pp_runs_label(const PpRunsBatch& B, int npages, int want, hipStream_t st) {
    int hmax = 1;
    for (int i = 0; i < npages; ++i) hmax = B.h[i] > hmax ? B.h[i] : hmax;
    const int rows_per_wg = kPpThreads / 64, wgs = (hmax + rows_per_wg - 1) / rows_per_wg;
    hipLaunchKernelGGL(pp_runs_count_kernel, dim3(wgs, npages), dim3(kPpThreads), 0, st, B, want);
    hipLaunchKernelGGL(pp_runs_scan_kernel, dim3(1, npages), dim3(1024), 0, st, B);
    hipLaunchKernelGGL(pp_runs_write_kernel, dim3(wgs, npages), dim3(kPpThreads), 0, st, B, want);
    hipLaunchKernelGGL(pp_runs_union_kernel, dim3((hmax + kRunBand - 1) / kRunBand, npages), dim3(kPpThreads), 0, st, B);
    if (hmax > kRunBand) hipLaunchKernelGGL(pp_runs_union_borders_kernel, dim3((hmax - 1) / kRunBand, npages), dim3(kPpThreads), 0, st, B);
    hipLaunchKernelGGL(pp_runs_stats_kernel, dim3(kRunBlocks, npages), dim3(kPpThreads), 0, st, B);
}


// ---- the kernels above as launches: one image (the single-page entry points) ...
__global__ __launch_bounds__(kPpThreads) void pp_hist_kernel(const uint8_t* img, int64_t n, uint32_t* hist) { pp_hist_kernel_body(img, n, hist); }
__global__ __launch_bounds__(kPpThreads) void pp_threshold_kernel(const uint8_t* img, int64_t n, int thr, int invert, uint8_t* ink) { pp_threshold_kernel_body(img, n, thr, invert, ink); }
__global__ __launch_bounds__(kPpThreads) void pp_ink_points_kernel(const uint8_t* ink, int h, int w, int step, uint32_t* points, uint32_t* count) { pp_ink_points_kernel_body(ink, h, w, step, points, count); }
__global__ __launch_bounds__(kPpThreads) void pp_angle_hist_points_kernel(const uint32_t* points, const uint32_t* count, int hs, int wsm, const double* cs, uint32_t* hist) { pp_angle_hist_points_kernel_body(points, count, hs, wsm, cs, hist); }
__global__ __launch_bounds__(kPpThreads) void pp_rotate_kernel(const uint8_t* ink, int h, int w, uint8_t* out, int oh, int ow, const double* mo) { pp_rotate_kernel_body(ink, h, w, out, oh, ow, mo); }
__global__ __launch_bounds__(kPpThreads) void pp_open_runs_kernel(const uint8_t* in, uint8_t* out, int h, int w, int len, int axis) { pp_open_runs_kernel_body(in, out, h, w, len, axis); }
__global__ __launch_bounds__(kPpThreads) void pp_row_sums_kernel(const uint8_t* ink, int h, int w, int32_t* sums) { pp_row_sums_kernel_body(ink, h, w, sums); }
__global__ __launch_bounds__(kPpThreads) void pp_cut_strips_kernel(const uint8_t* __restrict__ ink, int w, const int64_t* __restrict__ boxes, uint8_t* __restrict__ out) { pp_cut_strips_kernel_body(ink, w, boxes, out); }
__global__ __launch_bounds__(kPpThreads) void pp_clear_rows_kernel(uint8_t* ink, int w, const int32_t* rows, int nrows) { pp_clear_rows_kernel_body(ink, w, rows, nrows); }

// ... and the pages of a stage call in ONE launch (blockIdx.z = page; grids sized for the largest page, every body is
// grid-stride or returns for what lies beyond its page): in a stream a batch's chain is as long as its launches are many
struct PpPages {
    const void* a[kRunPages]; void* b[kRunPages]; const void* c[kRunPages]; void* d[kRunPages];
    int h[kRunPages], w[kRunPages], i0[kRunPages], i1[kRunPages];
    long long n[kRunPages];
};
__global__ __launch_bounds__(kPpThreads) void pp_hist_pages_kernel(PpPages B) {
    const int p = blockIdx.z;
    pp_hist_kernel_body((const uint8_t*)B.a[p], B.n[p], (uint32_t*)B.b[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_threshold_pages_kernel(PpPages B) {
    const int p = blockIdx.z;
    pp_threshold_kernel_body((const uint8_t*)B.a[p], B.n[p], B.i0[p], 0, (uint8_t*)B.b[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_ink_points_pages_kernel(PpPages B) {
    const int p = blockIdx.z;
    pp_ink_points_kernel_body((const uint8_t*)B.a[p], B.h[p], B.w[p], B.i0[p], (uint32_t*)B.b[p], (uint32_t*)B.d[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_angle_hist_points_pages_kernel(PpPages B) {      // i0 = hs, i1 = ws, n = angles
    const int p = blockIdx.z;
    if ((long long)blockIdx.y >= B.n[p] || !B.i0[p] || !B.i1[p]) return;
    pp_angle_hist_points_kernel_body((const uint32_t*)B.a[p], (const uint32_t*)B.c[p], B.i0[p], B.i1[p], (const double*)B.d[p],
                                     (uint32_t*)B.b[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_rotate_pages_kernel(PpPages B) {                 // i0 = oh, i1 = ow, c = map or NULL
    const int p = blockIdx.z;
    if (B.c[p]) { pp_rotate_kernel_body((const uint8_t*)B.a[p], B.h[p], B.w[p], (uint8_t*)B.b[p], B.i0[p], B.i1[p], (const double*)B.c[p]); return; }
    const uint8_t* src = (const uint8_t*)B.a[p];
    uint8_t* dst = (uint8_t*)B.b[p];
    const int64_t n = (int64_t)B.i0[p] * B.i1[p];                                                 // (not turned: a copy)
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) dst[e] = src[e];
}
__global__ __launch_bounds__(kPpThreads) void pp_open_runs_pages_kernel(PpPages B, int len, int axis) {
    const int p = blockIdx.z;
    pp_open_runs_kernel_body((const uint8_t*)B.a[p], (uint8_t*)B.b[p], B.h[p], B.w[p], len, axis);
}
__global__ __launch_bounds__(kPpThreads) void pp_copy_pages_kernel(PpPages B) {
    const int p = blockIdx.z;
    const uint8_t* src = (const uint8_t*)B.a[p];
    uint8_t* dst = (uint8_t*)B.b[p];
    const int64_t n = B.n[p];
    for (int64_t e = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; e < n; e += (int64_t)gridDim.x * kPpThreads) dst[e] = src[e];
}
__global__ __launch_bounds__(kPpThreads) void pp_row_sums_pages_kernel(PpPages B) {
    const int p = blockIdx.z;
    if ((int)blockIdx.x >= B.h[p]) return;
    pp_row_sums_kernel_body((const uint8_t*)B.a[p], B.h[p], B.w[p], (int32_t*)B.b[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_clear_rows_pages_kernel(PpPages B) {             // c = rows, i0 = their number
    const int p = blockIdx.z;
    if ((int)blockIdx.x >= B.i0[p]) return;
    pp_clear_rows_kernel_body((uint8_t*)B.b[p], B.w[p], (const int32_t*)B.c[p], B.i0[p]);
}
__global__ __launch_bounds__(kPpThreads) void pp_cut_strips_pages_kernel(PpPages B, uint8_t* out) {   // c = boxes, i0 = their number
    const int p = blockIdx.z;
    if ((int)blockIdx.y >= B.i0[p]) return;
    pp_cut_strips_kernel_body((const uint8_t*)B.a[p], B.w[p], (const int64_t*)B.c[p], out);
}

static int pp_blocks(int64_t n) {
    const int64_t b = (n + kPpThreads - 1) / kPpThreads;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
// the byte-plane kernels that take 16 bytes per lane: a lane per 16 bytes (and enough for the scalar tail / fallback)
static int pp_blocks16(int64_t n) {
    const int64_t b = ((n + 15) / 16 + kPpThreads - 1) / kPpThreads;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace ta

using namespace ta;

#define PP_LAUNCH_CHECK(what) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return ta_fail_hip(e_, what); } while (0)

extern "C" int ta_pp_histogram(const uint8_t* img, int64_t n, uint32_t* hist256, void* stream) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!img || !hist256) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(uint32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "histogram memset");
    if (n) hipLaunchKernelGGL(pp_hist_kernel, dim3(pp_blocks16(n) > 1024 ? 1024 : pp_blocks16(n)), dim3(kPpThreads), 0, st, img, n, hist256);
    PP_LAUNCH_CHECK("pp_hist_kernel");
    return TA_OK;
}

extern "C" int ta_pp_threshold(const uint8_t* img, int64_t n, int32_t thr, int32_t invert, uint8_t* ink, void* stream) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!img || !ink) return ta_fail(TA_EINVAL, "null pointer argument");
    if (n) hipLaunchKernelGGL(pp_threshold_kernel, dim3(pp_blocks16(n)), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), img, n, thr, invert, ink);
    PP_LAUNCH_CHECK("pp_threshold_kernel");
    return TA_OK;
}

// 8-connected components of `ink` (h x w): lab[p] = linear index of the component's first pixel,
// -1 on background.  stats: five int32 arrays of h*w entries (area, x0, y0, x1, y1; indexed by the
// root).  flag: one device int, unused (nothing iterates any more).  Asynchronous on `stream`.
extern "C" int ta_pp_label(const uint8_t* ink, int32_t h, int32_t w, int32_t* lab, int32_t* stats,
                           int32_t* flag, void* stream) {
    if (h < 0 || w < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!ink || !lab || !stats || !flag) return ta_fail(TA_EINVAL, "null pointer argument");
    const int64_t n = (int64_t)h * w;
    if (n == 0) return TA_OK;
    if (n >= (1ll << 31)) return ta_fail(TA_ELIMIT, "page too large for 32-bit labels");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = pp_blocks(n);
    hipLaunchKernelGGL(pp_label_tile_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH),
                       dim3(kTileH * kTileW), 0, st, ink, h, w, lab);
    // stitching across tiles (one union-find pass over the tile borders) and path compression
    hipLaunchKernelGGL(pp_label_merge_kernel, dim3(nb), dim3(kPpThreads), 0, st, lab, h, w);
    hipLaunchKernelGGL(pp_label_flatten_kernel, dim3(nb), dim3(kPpThreads), 0, st, lab, n, stats);
    (void)flag;
    int32_t* area = stats; int32_t* x0 = stats + n; int32_t* y0 = stats + 2 * n;
    int32_t* x1 = stats + 3 * n; int32_t* y1 = stats + 4 * n;
    hipLaunchKernelGGL(pp_stats_kernel, dim3(nb > kStatBlocks ? kStatBlocks : nb), dim3(kPpThreads), 0, st, lab, h, w, area, x0, y0, x1, y1);
    PP_LAUNCH_CHECK("pp_label kernels");
    return TA_OK;
}

// The same for `nimg` images at once: ink / lab / stats are HOST arrays of device pointers, h / w host
// arrays (flags: unused since the labelling needs no convergence loop; kept in the signature).
extern "C" int ta_pp_label_batch(int32_t nimg, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                                 int32_t* const* lab, int32_t* const* stats, int32_t* flags, void* stream) {
    if (nimg < 0) return ta_fail(TA_EINVAL, "negative size");
    if (nimg == 0) return TA_OK;
    if (!ink || !h || !w || !lab || !stats || !flags) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    std::vector<char> active(nimg, 0);
    for (int i = 0; i < nimg; ++i) {
        if (h[i] < 0 || w[i] < 0) return ta_fail(TA_EINVAL, "negative size");
        const int64_t n = (int64_t)h[i] * w[i];
        if (n == 0) continue;
        if (n >= (1ll << 31)) return ta_fail(TA_ELIMIT, "page too large for 32-bit labels");
        if (!ink[i] || !lab[i] || !stats[i]) return ta_fail(TA_EINVAL, "null pointer argument");
        active[i] = 1;
        hipLaunchKernelGGL(pp_label_tile_kernel, dim3((w[i] + kTileW - 1) / kTileW, (h[i] + kTileH - 1) / kTileH),
                           dim3(kTileH * kTileW), 0, st, ink[i], h[i], w[i], lab[i]);
    }
    for (int i = 0; i < nimg; ++i) {
        if (!active[i]) continue;
        const int64_t n = (int64_t)h[i] * w[i];
        const int nb = pp_blocks(n);
        hipLaunchKernelGGL(pp_label_merge_kernel, dim3(nb), dim3(kPpThreads), 0, st, lab[i], h[i], w[i]);
        hipLaunchKernelGGL(pp_label_flatten_kernel, dim3(nb), dim3(kPpThreads), 0, st, lab[i], n, stats[i]);
    }
    (void)flags;
    for (int i = 0; i < nimg; ++i) {
        const int64_t n = (int64_t)h[i] * w[i];
        if (n == 0) continue;
        const int nb = pp_blocks(n);
        int32_t* area = stats[i]; int32_t* x0 = area + n; int32_t* y0 = area + 2 * n;
        int32_t* x1 = area + 3 * n; int32_t* y1 = area + 4 * n;
        hipLaunchKernelGGL(pp_stats_kernel, dim3(nb > kStatBlocks ? kStatBlocks : nb), dim3(kPpThreads), 0, st, lab[i], h[i], w[i], area, x0, y0, x1, y1);
    }
    PP_LAUNCH_CHECK("pp_label_batch kernels");
    return TA_OK;
}

// component table: up to `cap` records {root, area, x0, y0, x1, y1}; *count receives the true number
extern "C" int ta_pp_components(const int32_t* lab, const int32_t* stats, int32_t h, int32_t w,
                                int32_t* recs, int32_t cap, int32_t* count, void* stream) {
    if (h < 0 || w < 0 || cap < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!lab || !stats || !recs || !count) return ta_fail(TA_EINVAL, "null pointer argument");
    const int64_t n = (int64_t)h * w;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "component count memset");
    if (n) hipLaunchKernelGGL(pp_collect_kernel, dim3(pp_blocks(n)), dim3(kPpThreads), 0, st, lab, n, stats,
                              stats + n, stats + 2 * n, stats + 3 * n, stats + 4 * n, recs, cap, count);
    PP_LAUNCH_CHECK("pp_collect_kernel");
    return TA_OK;
}

// drop components with fewer than min_area pixels or taller than max_height rows
extern "C" int ta_pp_filter_components(uint8_t* ink, const int32_t* lab, const int32_t* stats, int32_t h,
                                       int32_t w, int32_t min_area, int32_t max_height, void* stream) {
    if (h < 0 || w < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!ink || !lab || !stats) return ta_fail(TA_EINVAL, "null pointer argument");
    const int64_t n = (int64_t)h * w;
    if (n) hipLaunchKernelGGL(pp_filter_kernel, dim3(pp_blocks(n)), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), ink, lab, n, stats, stats + 2 * n,
                              stats + 4 * n, min_area, max_height);
    PP_LAUNCH_CHECK("pp_filter_kernel");
    return TA_OK;
}

extern "C" int ta_pp_invert(uint8_t* ink, int64_t n, void* stream) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!ink) return ta_fail(TA_EINVAL, "null pointer argument");
    if (n) hipLaunchKernelGGL(pp_invert_kernel, dim3(pp_blocks16(n)), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), ink, n);
    PP_LAUNCH_CHECK("pp_invert_kernel");
    return TA_OK;
}

// hist[a][row], a < nang, row < ceil(h / step): row projection of the decimated page rotated by
// angle a; cos_sin = {cos a0, sin a0, cos a1, ...}
extern "C" int ta_pp_angle_histograms(const uint8_t* ink, int32_t h, int32_t w, int32_t step,
                                      const double* cos_sin, int32_t nang, uint32_t* hist, void* stream) {
    if (h < 0 || w < 0 || step < 1 || nang < 0) return ta_fail(TA_EINVAL, "bad size");
    if (!ink || !cos_sin || !hist) return ta_fail(TA_EINVAL, "null pointer argument");
    const int hs = (h + step - 1) / step, wsm = (w + step - 1) / step;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(uint32_t) * (size_t)nang * hs, st);
    if (e != hipSuccess) return ta_fail_hip(e, "angle histogram memset");
    const int64_t n = (int64_t)hs * wsm;
    // pixel slices per angle: enough workgroups to fill the chip, few enough that the per-workgroup flush
    // (hs atomics) stays small beside the slice's own work
    int slices = 512 / (nang > 0 ? nang : 1);
    slices = slices < 1 ? 1 : (slices > 32 ? 32 : slices);
    const int64_t nitems = (int64_t)hs * ((wsm + kAngleRun - 1) / kAngleRun);
    if (slices > pp_blocks(nitems)) slices = pp_blocks(nitems) > 0 ? pp_blocks(nitems) : 1;
    if (n && nang) hipLaunchKernelGGL(pp_angle_hist_kernel, dim3(slices, nang), dim3(kPpThreads), 0, st, ink, h, w,
                                      step, cos_sin, nang, hist);
    PP_LAUNCH_CHECK("pp_angle_hist_kernel");
    return TA_OK;
}

// The same histograms from a list of the page's ink pixels, made once for all angles and sweeps:
// ta_pp_ink_points fills points (room for ceil(h / step) * ceil(w / step) entries) and *count [dev];
// ta_pp_angle_histograms_points(points, count, hs = ceil(h / step), ws = ceil(w / step), ...) = ta_pp_angle_histograms
extern "C" int ta_pp_ink_points(const uint8_t* ink, int32_t h, int32_t w, int32_t step, uint32_t* points,
                                uint32_t* count, void* stream) {
    if (h < 0 || w < 0 || step < 1) return ta_fail(TA_EINVAL, "bad size");
    if (!ink || !points || !count) return ta_fail(TA_EINVAL, "null pointer argument");
    const int hs = (h + step - 1) / step, wsm = (w + step - 1) / step;
    if (hs > 65535 || wsm > 65535) return ta_fail(TA_ELIMIT, "decimated page too large for 16-bit point coordinates");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "point count memset");
    const int64_t n = (int64_t)hs * wsm;
    if (n) hipLaunchKernelGGL(pp_ink_points_kernel, dim3(pp_blocks(n)), dim3(kPpThreads), 0, st, ink, h, w, step,
                              points, count);
    PP_LAUNCH_CHECK("pp_ink_points_kernel");
    return TA_OK;
}

extern "C" int ta_pp_angle_histograms_points(const uint32_t* points, const uint32_t* count, int32_t hs, int32_t ws,
                                             const double* cos_sin, int32_t nang, uint32_t* hist, void* stream) {
    if (hs < 0 || ws < 0 || nang < 0) return ta_fail(TA_EINVAL, "bad size");
    if (!points || !count || !cos_sin || !hist) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(uint32_t) * (size_t)nang * hs, st);
    if (e != hipSuccess) return ta_fail_hip(e, "angle histogram memset");
    int slices = 512 / (nang > 0 ? nang : 1);
    slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
    if (hs && ws && nang) hipLaunchKernelGGL(pp_angle_hist_points_kernel, dim3(slices, nang), dim3(kPpThreads), 0, st,
                                             points, count, hs, ws, cos_sin, hist);
    PP_LAUNCH_CHECK("pp_angle_hist_points_kernel");
    return TA_OK;
}

// mo = {m00, m01, m10, m11, offset0, offset1} of scipy.ndimage.rotate's affine map (output -> input)
extern "C" int ta_pp_rotate(const uint8_t* ink, int32_t h, int32_t w, uint8_t* out, int32_t oh, int32_t ow,
                            const double* mo, void* stream) {
    if (h < 0 || w < 0 || oh < 0 || ow < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!ink || !out || !mo) return ta_fail(TA_EINVAL, "null pointer argument");
    const int64_t n = (int64_t)oh * ow;
    if (n) hipLaunchKernelGGL(pp_rotate_kernel, dim3(pp_blocks(n)), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), ink, h, w, out, oh, ow, mo);
    PP_LAUNCH_CHECK("pp_rotate_kernel");
    return TA_OK;
}

extern "C" int ta_pp_open_runs(const uint8_t* in, uint8_t* out, int32_t h, int32_t w, int32_t len,
                               int32_t axis, void* stream) {
    if (h < 0 || w < 0 || len < 1 || (axis != 0 && axis != 1)) return ta_fail(TA_EINVAL, "bad argument");
    if (!in || !out) return ta_fail(TA_EINVAL, "null pointer argument");
    const int64_t n = (int64_t)h * w;
    if (n) hipLaunchKernelGGL(pp_open_runs_kernel, dim3(pp_blocks(n)), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), in, out, h, w, len, axis);
    PP_LAUNCH_CHECK("pp_open_runs_kernel");
    return TA_OK;
}

extern "C" int ta_pp_row_sums(const uint8_t* ink, int32_t h, int32_t w, int32_t* sums, void* stream) {
    if (h < 0 || w < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!ink || !sums) return ta_fail(TA_EINVAL, "null pointer argument");
    if (h) hipLaunchKernelGGL(pp_row_sums_kernel, dim3(h), dim3(kPpThreads), 0,
                              reinterpret_cast<hipStream_t>(stream), ink, h, w, sums);
    PP_LAUNCH_CHECK("pp_row_sums_kernel");
    return TA_OK;
}

// Host arithmetic (no device work): the arguments of the logarithms in calculate_peak_prominence
// (reference textAlignPreprocessing.py:59-110) for the candidate rows idx[0..k) of a projection d[0..n) --
// rows that passed its local-maximum test.  arg[c] = d[i] where d[i] is the maximum, else
// d[i] - min(d[lo:hi]) + 1 with [lo, hi) running from the nearest strictly higher sample (the left one
// only if it is strictly nearer) to i, as the reference slices it.  Plain float64, evaluated left to right.
static inline double pp_peak_arg(const double* d, int n, int i, double data_max) {
    const double here = d[i];
    if (here == data_max) return here;
    int nr = i + 1, nl = i - 1;
    while (nr < n && !(d[nr] > here)) ++nr;
    while (nl >= 0 && !(d[nl] > here)) --nl;
    const bool has_r = nr < n, has_l = nl >= 0;
    if (!has_r && !has_l) return here;                               // (data_max was not the maximum: as if it were)
    const bool go_left = has_l && (!has_r || (nr - i) > (i - nl));
    const int lo = go_left ? nl : i, hi = go_left ? i : nr;
    double key = d[lo];
    for (int j = lo + 1; j < hi; ++j) key = d[j] < key ? d[j] : key;
    return here - key + 1.0;
}

extern "C" int ta_pp_peak_prominence_args(const double* d, int32_t n, const int32_t* idx, int32_t k,
                                          double data_max, double* arg) {
    if (n < 0 || k < 0) return ta_fail(TA_EINVAL, "negative size");
    if (k == 0) return TA_OK;
    if (!d || !idx || !arg) return ta_fail(TA_EINVAL, "null pointer argument");
    for (int c = 0; c < k; ++c) {
        const int i = idx[c];
        if (i < 0 || i >= n) return ta_fail(TA_EINVAL, "candidate row outside the projection");
        arg[c] = pp_peak_arg(d, n, i, data_max);
    }
    return TA_OK;
}

// The text-line peaks of a batch of pages, host arithmetic in two calls with ONE numpy logarithm in between (numpy's
// float64 log is its own SIMD routine where the CPU has AVX-512 and differs from libm's in the last bit now and then;
// the prominences are compared with a tolerance after a division, so the logarithm stays numpy's):
//
// ta_host_peak_candidates: page k's row projection proj + off[k], len[k] int64 sums -> smoothed + off[k] (float64;
//   moving_avg_filter, reference :147-157: mean over filter_size rows to either side, the ends zero; window sums of
//   integers are exact, then one division), its local maxima (the test calculate_peak_prominence starts with, :59-72)
//   cand_idx / cand_arg at cand_off[k] .. + cand_n[k] (cand_off[k] = off[k]: a page has fewer candidates than rows) with
//   the ARGUMENT of each one's logarithm (ta_pp_peak_prominence_args).
// ta_host_peak_select: with cand_log = np.log(cand_arg): per page the candidates whose prominence / the largest
//   exceeds tol (find_peak_locations :113-144, including its removal of the first of two equal neighbours among all but
//   the last pair), peaks + off[k] (npeaks[k] of them); and the white rows between neighbouring peaks (:222-232):
//   for each pair the first minimum of the smoothed projection between them and the row above it, sorted, without
//   repeats, rows + 2 * off[k] (nrows[k] of them).
extern "C" int ta_host_peak_candidates(const int64_t* proj, const int64_t* off, const int32_t* len, int32_t n,
                                       int32_t filter_size, double* smoothed, int32_t* cand_idx, double* cand_arg,
                                       int32_t* cand_n) {
    if (n < 0 || filter_size < 0) return ta_fail(TA_EINVAL, "negative size");
    if (n == 0) return TA_OK;
    if (!proj || !off || !len || !smoothed || !cand_idx || !cand_arg || !cand_n) return ta_fail(TA_EINVAL, "null pointer argument");
    const int fs = filter_size;
    for (int32_t k = 0; k < n; ++k) {
        if (len[k] < 0 || off[k] < 0) return ta_fail(TA_EINVAL, "negative size");
        const int m = len[k];
        const int64_t* p = proj + off[k];
        double* sm = smoothed + off[k];
        for (int i = 0; i < m; ++i) sm[i] = 0.0;
        if (m > 2 * fs) {
            int64_t win = 0;
            for (int i = 0; i < 2 * fs + 1; ++i) win += p[i];
            const double width = (double)(2 * fs + 1);
            for (int i = fs; i < m - fs; ++i) {
                sm[i] = (double)win / width;
                if (i + fs + 1 < m) win += p[i + fs + 1] - p[i - fs];
            }
        }
        double data_max = m ? sm[0] : 0.0;
        for (int i = 1; i < m; ++i) data_max = sm[i] > data_max ? sm[i] : data_max;
        int c = 0;
        for (int i = 1; i + 1 < m; ++i) {
            const double l = sm[i - 1], mid = sm[i], r = sm[i + 1];
            if (l > mid || r > mid || (l == mid && r == mid)) continue;
            cand_idx[off[k] + c] = i;
            cand_arg[off[k] + c] = pp_peak_arg(sm, m, i, data_max);
            ++c;
        }
        cand_n[k] = c;
    }
    return TA_OK;
}

extern "C" int ta_host_peak_select(const double* smoothed, const int64_t* off, const int32_t* len, int32_t n,
                                   const int32_t* cand_idx, const double* cand_log, const int32_t* cand_n, double tol,
                                   int32_t* peaks, int32_t* npeaks, int32_t* rows, int32_t* nrows) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (n == 0) return TA_OK;
    if (!smoothed || !off || !len || !cand_idx || !cand_log || !cand_n || !peaks || !npeaks || !rows || !nrows)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if (!(tol >= 0.0)) return ta_fail(TA_EINVAL, "a negative tolerance makes every row a peak: the host language's loop covers it");
    std::vector<int> idx;
    std::vector<double> val;
    for (int32_t k = 0; k < n; ++k) {
        const int c = cand_n[k];
        if (c < 0 || c > len[k]) return ta_fail(TA_EINVAL, "bad candidate count");
        const int32_t* ci = cand_idx + off[k];
        const double* cl = cand_log + off[k];
        const double* sm = smoothed + off[k];
        npeaks[k] = 0; nrows[k] = 0;
        double top = 0.0;                                             // max(vals + [0])
        for (int j = 0; j < c; ++j) top = cl[j] > top ? cl[j] : top;
        if (top == 0.0) continue;
        idx.clear(); val.clear();
        for (int j = 0; j < c; ++j) {
            const double v = cl[j] / top;
            if (v > tol) { idx.push_back(ci[j]); val.push_back(v); }
        }
        // both corners of a flat-topped peak are prominent: the first of two equal neighbours goes (the reference looks
        // at all pairs but the last one)
        const int np0 = (int)idx.size();
        int np = 0;
        int32_t* out = peaks + off[k];
        for (int j = 0; j < np0; ++j) {
            if (j < np0 - 2 && val[j] == val[j + 1]) continue;
            if (idx[j] < 0 || idx[j] >= len[k]) return ta_fail(TA_EINVAL, "candidate row outside the projection");
            out[np++] = idx[j];
        }
        npeaks[k] = np;
        int32_t* rw = rows + 2 * off[k];
        int nr = 0;
        for (int j = 0; j + 1 < np; ++j) {
            const int a = out[j], b = out[j + 1];
            int at = a;
            for (int i = a + 1; i < b; ++i) if (sm[i] < sm[at]) at = i;     // np.argmin: the first minimum
            const int lo = at - 1 > 0 ? at - 1 : 0;
            // (peaks ascend, so the rows do: appending while skipping repeats keeps them sorted and unique)
            if (nr == 0 || rw[nr - 1] < lo) rw[nr++] = lo;
            if (rw[nr - 1] < at) rw[nr++] = at;
        }
        nrows[k] = nr;
    }
    return TA_OK;
}

extern "C" int ta_pp_cut_strips(const uint8_t* ink, int32_t h, int32_t w, const int64_t* boxes, int32_t nstrips,
                                uint8_t* out, void* stream) {
    if (h < 0 || w < 0 || nstrips < 0) return ta_fail(TA_EINVAL, "negative size");
    if (nstrips == 0) return TA_OK;
    if (!ink || !boxes || !out) return ta_fail(TA_EINVAL, "null pointer argument");
    hipLaunchKernelGGL(pp_cut_strips_kernel, dim3(64, nstrips), dim3(kPpThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), ink, w, boxes, out);
    PP_LAUNCH_CHECK("pp_cut_strips_kernel");
    return TA_OK;
}

extern "C" int ta_pp_clear_rows(uint8_t* ink, int32_t w, const int32_t* rows, int32_t nrows, void* stream) {
    if (w < 0 || nrows < 0) return ta_fail(TA_EINVAL, "negative size");
    if (nrows == 0) return TA_OK;
    if (!ink || !rows) return ta_fail(TA_EINVAL, "null pointer argument");
    hipLaunchKernelGGL(pp_clear_rows_kernel, dim3(nrows), dim3(kPpThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), ink, w, rows, nrows);
    PP_LAUNCH_CHECK("pp_clear_rows_kernel");
    return TA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole STAGES of the preprocessing for a batch of pages, one call each: everything between two of the pipeline's
// data-dependent host decisions (Otsu threshold | skew sweeps | projection peaks | component selection | strips).
// A page's stage is a dozen launches; made one by one from the host language they cost more host time than the kernels
// take (and a Python host holds its interpreter lock meanwhile, so the page threads could not run beside each other).
// All pointer arguments named like arrays are HOST arrays (n entries) of DEVICE pointers / sizes; everything is
// enqueued on `stream`, nothing is waited for.  Same kernels, same order per page as the single-page entry points.
namespace {

int pp_check_pages(int32_t n, const int32_t* h, const int32_t* w) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (n && (!h || !w)) return ta_fail(TA_EINVAL, "null pointer argument");
    for (int i = 0; i < n; ++i) {
        if (h[i] < 0 || w[i] < 0) return ta_fail(TA_EINVAL, "negative size");
        if ((int64_t)h[i] * w[i] >= (1ll << 31)) return ta_fail(TA_ELIMIT, "page too large for 32-bit labels");
    }
    return TA_OK;
}

}  // namespace

extern "C" int ta_pp_histogram_batch(int32_t n, const uint8_t* const* img, const int64_t* npix, uint32_t* hist,
                                     void* stream) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (n == 0) return TA_OK;
    if (!img || !npix || !hist) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)n * 256 * sizeof(uint32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "histogram memset");
    for (int i = 0; i < n; ++i) {
        if (npix[i] < 0) return ta_fail(TA_EINVAL, "negative size");
        if (npix[i] && !img[i]) return ta_fail(TA_EINVAL, "null pointer argument");
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages B;
        int m = 0;
        int64_t big = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            if (!npix[i]) continue;
            B.a[m] = img[i]; B.n[m] = npix[i]; B.b[m] = hist + (size_t)i * 256;
            big = npix[i] > big ? npix[i] : big;
            ++m;
        }
        if (m) hipLaunchKernelGGL(pp_hist_pages_kernel, dim3(pp_blocks16(big) > 1024 ? 1024 : pp_blocks16(big), 1, m), dim3(kPpThreads), 0, st, B);
    }
    PP_LAUNCH_CHECK("pp_hist_kernel");
    return TA_OK;
}

// greyscale pages -> cleaned ink planes + the ink points of the decimated pages (reference
// textAlignPreprocessing.py:167-186): threshold at thr[i]; despeckle (components under `despeckle` pixels) the ink,
// then the background (small holes), drop components taller than max_height rows; list the ink pixels of the page
// decimated by step[i] (points[i], counts[i]: ta_pp_ink_points).  lab[i] / stats[i]: h*w and 5*h*w int32 of scratch.
extern "C" int ta_pp_binarise_batch(int32_t n, const uint8_t* const* img, const int32_t* h, const int32_t* w,
                                    const int32_t* thr, int32_t despeckle, int32_t max_height, uint8_t* const* ink,
                                    int32_t* const* lab, int32_t* const* stats, const int32_t* step,
                                    uint32_t* const* points, uint32_t* counts, int32_t flags, void* stream) {
    int rc = pp_check_pages(n, h, w);
    if (rc != TA_OK) return rc;
    if (n == 0) return TA_OK;
    if (!img || !thr || !ink || !lab || !stats || !step || !points || !counts) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool by_runs = !(flags & TA_PP_LABEL_PIXELS);
    for (int i = 0; i < n; ++i) {
        const int64_t np = (int64_t)h[i] * w[i];
        if (step[i] < 1) return ta_fail(TA_EINVAL, "bad size");
        if (np && (!img[i] || !ink[i] || !lab[i] || !stats[i] || !points[i])) return ta_fail(TA_EINVAL, "null pointer argument");
        const int hs = (h[i] + step[i] - 1) / step[i], wsm = (w[i] + step[i] - 1) / step[i];
        if (hs > 65535 || wsm > 65535) return ta_fail(TA_ELIMIT, "decimated page too large for 16-bit point coordinates");
        if (np && !pp_runs_fit(h[i], w[i])) by_runs = false;
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages B;
        int m = 0;
        int64_t big = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            const int64_t np = (int64_t)h[i] * w[i];
            if (!np) continue;
            B.a[m] = img[i]; B.n[m] = np; B.i0[m] = thr[i]; B.b[m] = ink[i];
            big = np > big ? np : big;
            ++m;
        }
        if (m) hipLaunchKernelGGL(pp_threshold_pages_kernel, dim3(pp_blocks16(big), 1, m), dim3(kPpThreads), 0, st, B);
    }
    if (by_runs) {
        // three labellings over runs: ink specks out, paper specks (holes) in -- the runs of PAPER are labelled, no
        // inversion of the plane and back --, tall components out; kRunPages pages per launch
        for (int p0 = 0; p0 < n; p0 += kRunPages) {
            PpRunsBatch B;
            int m = 0;
            for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
                if (!((int64_t)h[i] * w[i])) continue;
                B.ink[m] = ink[i]; B.h[m] = h[i]; B.w[m] = w[i]; B.R[m] = pp_runs_in(lab[i], stats[i], h[i], w[i]);
                ++m;
            }
            if (!m) continue;
            for (int round = 0; round < 3; ++round) {
                const int want = round == 1 ? 0 : 1;
                pp_runs_label(B, m, want, st);
                hipLaunchKernelGGL(pp_runs_filter_kernel, dim3(kRunBlocks, m), dim3(kPpThreads), 0, st, B,
                                   round < 2 ? despeckle : 0, round < 2 ? (1 << 30) : max_height, want ? 0 : 1);
            }
        }
    } else {
        for (int round = 0; round < 3; ++round) {
            rc = ta_pp_label_batch(n, ink, h, w, lab, stats, reinterpret_cast<int32_t*>(counts), stream);
            if (rc != TA_OK) return rc;
            for (int i = 0; i < n; ++i) {
                const int64_t np = (int64_t)h[i] * w[i];
                if (!np) continue;
                hipLaunchKernelGGL(pp_filter_kernel, dim3(pp_blocks(np)), dim3(kPpThreads), 0, st, ink[i], lab[i], np, stats[i],
                                   stats[i] + 2 * np, stats[i] + 4 * np, round < 2 ? despeckle : 0, round < 2 ? (1 << 30) : max_height);
                // round 0 works on the ink, round 1 on the background (inverted before and after), round 2 on the ink again
                if (round < 2) hipLaunchKernelGGL(pp_invert_kernel, dim3(pp_blocks16(np)), dim3(kPpThreads), 0, st, ink[i], np);
            }
        }
    }
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n * sizeof(uint32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "point count memset");
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages B;
        int m = 0;
        int64_t big = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            const int hs = (h[i] + step[i] - 1) / step[i], wsm = (w[i] + step[i] - 1) / step[i];
            const int64_t np = (int64_t)hs * wsm;
            if (!np) continue;
            B.a[m] = ink[i]; B.h[m] = h[i]; B.w[m] = w[i]; B.i0[m] = step[i]; B.b[m] = points[i]; B.d[m] = counts + i;
            big = np > big ? np : big;
            ++m;
        }
        if (m) hipLaunchKernelGGL(pp_ink_points_pages_kernel, dim3(pp_blocks(big), 1, m), dim3(kPpThreads), 0, st, B);
    }
    PP_LAUNCH_CHECK("binarise stage kernels");
    return TA_OK;
}

// ta_pp_angle_histograms_points for every page: hist[i] = [nang[i]][hs[i]] uint32, cos_sin[i] = 2 * nang[i] doubles [dev]
extern "C" int ta_pp_angle_histograms_points_batch(int32_t n, const uint32_t* const* points, const uint32_t* counts,
                                                   const int32_t* hs, const int32_t* ws, const double* const* cos_sin,
                                                   const int32_t* nang, uint32_t* const* hist, void* stream) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative size");
    if (n == 0) return TA_OK;
    if (!points || !counts || !hs || !ws || !cos_sin || !nang || !hist) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        if (hs[i] < 0 || ws[i] < 0 || nang[i] < 0) return ta_fail(TA_EINVAL, "bad size");
        if (!points[i] || !cos_sin[i] || !hist[i]) return ta_fail(TA_EINVAL, "null pointer argument");
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages B;
        int m = 0, amax = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            if (!nang[i] || !hs[i]) continue;
            hipError_t e = hipMemsetAsync(hist[i], 0, sizeof(uint32_t) * (size_t)nang[i] * hs[i], st);
            if (e != hipSuccess) return ta_fail_hip(e, "angle histogram memset");
            B.a[m] = points[i]; B.c[m] = counts + i; B.i0[m] = hs[i]; B.i1[m] = ws[i]; B.d[m] = const_cast<double*>(cos_sin[i]);
            B.n[m] = nang[i]; B.b[m] = hist[i];
            amax = nang[i] > amax ? nang[i] : amax;
            ++m;
        }
        if (!m) continue;
        int slices = 512 / amax;
        slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
        hipLaunchKernelGGL(pp_angle_hist_points_pages_kernel, dim3(slices, amax, m), dim3(kPpThreads), 0, st, B);
    }
    PP_LAUNCH_CHECK("pp_angle_hist_points_kernel");
    return TA_OK;
}

// deskew + run filters + row projection (reference :187-195, :212-215): out[i] = ink[i] rotated through mo[i]
// (ta_pp_rotate; mo[i] NULL: a copy, oh / ow = h / w), eroded[i] = out[i] opened with runs of `runs_len` pixels along
// the rows then the columns, `rounds` times (tmp[i]: oh*ow bytes of scratch), sums[i][r] = ink pixels of eroded row r
extern "C" int ta_pp_deskew_batch(int32_t n, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                                  const double* const* mo, uint8_t* const* out, const int32_t* oh, const int32_t* ow,
                                  uint8_t* const* tmp, uint8_t* const* eroded, int32_t runs_len, int32_t rounds,
                                  int32_t* const* sums, void* stream) {
    int rc = pp_check_pages(n, h, w);
    if (rc == TA_OK) rc = pp_check_pages(n, oh, ow);
    if (rc != TA_OK) return rc;
    if (n == 0) return TA_OK;
    if (!ink || !mo || !out || !tmp || !eroded || !sums || rounds < 0) return ta_fail(TA_EINVAL, "bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        const int64_t np = (int64_t)oh[i] * ow[i];
        if (!np) continue;
        if (!ink[i] || !out[i] || !tmp[i] || !eroded[i] || !sums[i]) return ta_fail(TA_EINVAL, "null pointer argument");
        if (!mo[i] && (oh[i] != h[i] || ow[i] != w[i])) return ta_fail(TA_EINVAL, "a page that is not rotated keeps its size");
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages R, A, C, S;                                  // rotation | opening along the rows | along the columns | row sums
        int m = 0, hmax = 0;
        int64_t big = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            const int64_t np = (int64_t)oh[i] * ow[i];
            if (!np) continue;
            R.a[m] = ink[i]; R.h[m] = h[i]; R.w[m] = w[i]; R.b[m] = out[i]; R.i0[m] = oh[i]; R.i1[m] = ow[i]; R.c[m] = mo[i];
            A.a[m] = out[i]; A.b[m] = tmp[i]; A.h[m] = oh[i]; A.w[m] = ow[i];
            C.a[m] = tmp[i]; C.b[m] = eroded[i]; C.h[m] = oh[i]; C.w[m] = ow[i];
            S.a[m] = eroded[i]; S.h[m] = oh[i]; S.w[m] = ow[i]; S.b[m] = sums[i];
            big = np > big ? np : big;
            hmax = oh[i] > hmax ? oh[i] : hmax;
            ++m;
        }
        if (!m) continue;
        hipLaunchKernelGGL(pp_rotate_pages_kernel, dim3(pp_blocks(big), 1, m), dim3(kPpThreads), 0, st, R);
        bool opened = false;
        if (runs_len > 1) {
            for (int r = 0; r < rounds; ++r) {
                hipLaunchKernelGGL(pp_open_runs_pages_kernel, dim3(pp_blocks(big), 1, m), dim3(kPpThreads), 0, st, A, runs_len, 0);
                hipLaunchKernelGGL(pp_open_runs_pages_kernel, dim3(pp_blocks(big), 1, m), dim3(kPpThreads), 0, st, C, runs_len, 1);
                for (int k = 0; k < m; ++k) A.a[k] = C.b[k];      // a second round opens what the first left
                opened = true;
            }
        }
        if (!opened) {                                       // no filter: eroded = a copy of the plane
            PpPages K;
            for (int k = 0; k < m; ++k) { K.a[k] = R.b[k]; K.b[k] = C.b[k]; K.n[k] = (long long)R.i0[k] * R.i1[k]; }
            hipLaunchKernelGGL(pp_copy_pages_kernel, dim3(pp_blocks16(big), 1, m), dim3(kPpThreads), 0, st, K);
        }
        hipLaunchKernelGGL(pp_row_sums_pages_kernel, dim3(hmax, 1, m), dim3(kPpThreads), 0, st, S);
    }
    PP_LAUNCH_CHECK("deskew stage kernels");
    return TA_OK;
}

// the components of the text lines (reference :216-252): work[i] = eroded[i] with the rows rows[i][0..nrows[i]) cleared
// (the white lines between neighbouring text lines), labelled (lab[i] / stats[i] scratch as above), the component table
// of page i collected into recs + i*cap*6 (ta_pp_components), its true count into counts[i]
extern "C" int ta_pp_line_components_batch(int32_t n, const uint8_t* const* eroded, const int32_t* h, const int32_t* w,
                                           const int32_t* const* rows, const int32_t* nrows, uint8_t* const* work,
                                           int32_t* const* lab, int32_t* const* stats, int32_t* recs, int32_t cap,
                                           int32_t* counts, int32_t flags, void* stream) {
    int rc = pp_check_pages(n, h, w);
    if (rc != TA_OK) return rc;
    if (n == 0) return TA_OK;
    if (!eroded || !rows || !nrows || !work || !lab || !stats || !recs || !counts || cap < 0) return ta_fail(TA_EINVAL, "bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool by_runs = !(flags & TA_PP_LABEL_PIXELS);
    for (int i = 0; i < n; ++i) {
        const int64_t np = (int64_t)h[i] * w[i];
        if (!np) continue;
        if (!eroded[i] || !work[i] || !lab[i] || !stats[i] || nrows[i] < 0 || (nrows[i] && !rows[i])) return ta_fail(TA_EINVAL, "bad argument");
        if (!pp_runs_fit(h[i], w[i])) by_runs = false;
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages K;
        int m = 0, rmax = 0;
        int64_t big = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            const int64_t np = (int64_t)h[i] * w[i];
            if (!np) continue;
            K.a[m] = eroded[i]; K.b[m] = work[i]; K.n[m] = np; K.w[m] = w[i]; K.c[m] = rows[i]; K.i0[m] = nrows[i];
            big = np > big ? np : big;
            rmax = nrows[i] > rmax ? nrows[i] : rmax;
            ++m;
        }
        if (!m) continue;
        hipLaunchKernelGGL(pp_copy_pages_kernel, dim3(pp_blocks16(big), 1, m), dim3(kPpThreads), 0, st, K);
        if (rmax) hipLaunchKernelGGL(pp_clear_rows_pages_kernel, dim3(rmax, 1, m), dim3(kPpThreads), 0, st, K);
    }
    if (!by_runs) {
        rc = ta_pp_label_batch(n, work, h, w, lab, stats, counts, stream);
        if (rc != TA_OK) return rc;
    }
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n * sizeof(int32_t), st);
    if (e != hipSuccess) return ta_fail_hip(e, "component count memset");
    if (by_runs) {
        for (int p0 = 0; p0 < n; ) {
            // consecutive non-empty pages, kRunPages per launch (their tables are consecutive in recs / counts)
            while (p0 < n && !((int64_t)h[p0] * w[p0])) ++p0;
            if (p0 >= n) break;
            PpRunsBatch B;
            int m = 0;
            while (p0 + m < n && m < kRunPages && (int64_t)h[p0 + m] * w[p0 + m]) {
                const int i = p0 + m;
                B.ink[m] = work[i]; B.h[m] = h[i]; B.w[m] = w[i]; B.R[m] = pp_runs_in(lab[i], stats[i], h[i], w[i]);
                ++m;
            }
            pp_runs_label(B, m, 1, st);
            hipLaunchKernelGGL(pp_runs_collect_kernel, dim3(kRunBlocks, m), dim3(kPpThreads), 0, st, B,
                               recs + (size_t)p0 * cap * 6, cap, counts + p0);
            p0 += m;
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const int64_t np = (int64_t)h[i] * w[i];
            if (np) hipLaunchKernelGGL(pp_collect_kernel, dim3(pp_blocks(np)), dim3(kPpThreads), 0, st, lab[i], np, stats[i],
                                       stats[i] + np, stats[i] + 2 * np, stats[i] + 3 * np, stats[i] + 4 * np,
                                       recs + (size_t)i * cap * 6, cap, counts + i);
        }
    }
    PP_LAUNCH_CHECK("line component stage kernels");
    return TA_OK;
}

// ta_pp_cut_strips for every page, into ONE packed buffer (the boxes carry their offsets into it)
extern "C" int ta_pp_cut_strips_batch(int32_t n, const uint8_t* const* ink, const int32_t* h, const int32_t* w,
                                      const int64_t* const* boxes, const int32_t* nstrips, uint8_t* packed, void* stream) {
    int rc = pp_check_pages(n, h, w);
    if (rc != TA_OK) return rc;
    if (n == 0) return TA_OK;
    if (!ink || !boxes || !nstrips) return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        if (nstrips[i] < 0) return ta_fail(TA_EINVAL, "negative size");
        if (nstrips[i] && (!ink[i] || !boxes[i] || !packed)) return ta_fail(TA_EINVAL, "null pointer argument");
    }
    for (int p0 = 0; p0 < n; p0 += kRunPages) {
        PpPages B;
        int m = 0, smax = 0;
        for (int i = p0; i < n && i < p0 + kRunPages; ++i) {
            if (!nstrips[i]) continue;
            B.a[m] = ink[i]; B.w[m] = w[i]; B.c[m] = boxes[i]; B.i0[m] = nstrips[i];
            smax = nstrips[i] > smax ? nstrips[i] : smax;
            ++m;
        }
        if (m) hipLaunchKernelGGL(pp_cut_strips_pages_kernel, dim3(64, smax, m), dim3(kPpThreads), 0, st, B, packed);
    }
    PP_LAUNCH_CHECK("pp_cut_strips_kernel");
    return TA_OK;
}
