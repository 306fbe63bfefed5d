// ta_nw2.hip -- two-phase form of the affine-gap aligner (same results as ta_nw.hip, bit for bit).
//
// The single-pass fill kernel (ta_nw.hip) is bound by VALU issue, and 40 % of its instructions only
// serve the pointer byte (tag clean-up, two v_bfi, byte packing).  The reference needs pointers
// only along the traceback path (textSeqCompare.py:96-164), so:
//
//  phase 1  nw_score_kernel   the same strip / lane / skew wavefront on RAW integer scores: per cell
//           v_cmp + v_cndmask, 3 v_add, 3 v_max3_i32 and nothing else; no pointer byte is formed
//           or stored.  It leaves checkpoints in HBM (0.07 B per cell): every kCkGroups groups the
//           wave's whole lane state, and per strip its bottom row (XG, D) -- which is also what the
//           next strip starts from.
//  phase 2  nw_trace2_kernel  one wave per problem walks back strip by strip and, inside a strip,
//           checkpoint interval by checkpoint interval: it re-runs the TAGGED cell over the
//           interval the walk is in, restarted from that interval's state checkpoint (two halo
//           steps make the missing winner tags of the restart state irrelevant), keeps the
//           interval's pointer bytes in LDS and walks them.  The bottom rows carry no tags either:
//           a step that leaves a strip upwards gets its next state from the tagged outputs of the
//           cell it lands on, once the strip above is re-filled.  About 10 % of the cells are
//           recomputed.
//
// Data flow and the halo argument are replayed on the CPU by tests/native/sim_nw.cpp (run2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

#include "nw_cell.h"
#include "nw_hw.h"
#include "ta_common.h"

namespace ta {

#ifndef TA_CK_GROUPS
#define TA_CK_GROUPS 16
#endif
// State checkpoint every kCkGroups groups (a multiple of 4: the steady loop runs in blocks of 4 groups).
// The interval is also phase 2's chunk, whose pointer bytes (1 KiB per group) are what limits the
// number of traceback waves on a CU (measured by padding its LDS: 8 waves per CU 3.5 ms, 6: 4.8,
// 5: 5.6, 2: 10.7).  12 groups (10 waves per CU) were tried: 4.1 ms -- 28 % more chunks to set up and
// walk into, and 16 problems per CU do not divide into rounds of ten any better than into two of eight.
constexpr int kCkGroups = TA_CK_GROUPS;
static_assert(kCkGroups % 4 == 0 && kCkGroups >= 4, "whole blocks of four groups");
constexpr int kStateInts = 10;                // D[4], H[4], V[3], dsave
constexpr int kWsRange = 0x7ffff000;          // record count of a workspace buffer descriptor: every real offset is below

// Bottom rows kept per strip: not only the strip's last row (lane 63's, what the next strip starts from) but also
// lane 31's, so that phase 2 can restart HALF-strips of 128 rows (32 lanes) and carry two problems' half-strips in one
// wave (nw_trace2h_kernel).  The same two store instructions per group write both rows (each storing lane has its
// own offset); measured on phase 1 at 4096 x 4096^2: 11.30 ms with the strip rows only, 11.39 with lanes 31 and 63,
// 11.70 with lanes 15 / 31 / 47 / 63 (quarter-strips would save phase 2 no more than they cost here).
#ifndef TA_SUBROWS
#define TA_SUBROWS 2
#endif
constexpr int kSubRows = TA_SUBROWS;
constexpr int kSubLanes = 64 / kSubRows;
// per-problem layout of the phase-1/2 workspace (all offsets in bytes, 16-byte aligned)
struct Ws2 {
    int nstrips, ngroups, nck;
    int64_t rows, row_pitch, stck, total;
    __host__ __device__ Ws2(int n, int m) {
        using L = PtrLayout<4>;
        nstrips = L::nstrips(n);
        ngroups = L::ngroups(m);
        nck = ngroups / kCkGroups + 1;
        // bottom rows: row 0 is the table's boundary row, row 1 + kSubRows s + q the last row of lanes 16 q .. 16 q + 15
        // of strip s -- (XG or V~, D) per column, what the lane below consumes; top(s) = row kSubRows s is the row above
        // strip s.  Entry j sits at index j + 1, so that the four entries a group reads start on a 16-byte boundary.
        // Phase 1 hands a strip's results to the next strip through them (they are in L2 when the next wave, ~25
        // groups behind, reads them) and phase 2 restarts from them.  A lane that has passed its last column goes on
        // over pad columns (phase 1's steady loop): lane 15 writes entries up to column m + 51, hence the pitch.
        rows = 0;
        row_pitch = ((int64_t)(m + 72) * 8 + 15) & ~(int64_t)15;
        stck = rows + (int64_t)(nstrips * kSubRows + 1) * row_pitch;
        total = stck + (int64_t)nstrips * nck * kStateInts * 64 * 4;
    }
    __host__ __device__ int64_t row(int r) const { return rows + (int64_t)r * row_pitch; }
    __host__ __device__ int64_t top(int s) const { return row(s * kSubRows); }          // the row above strip s
    __host__ __device__ int64_t state(int s, int ck) const {
        return stck + ((int64_t)(s * nck + ck) * kStateInts * 64) * 4;
    }
};

struct RawRegs { int cmis, cmat, gox, goy; };

__device__ __forceinline__ void cell_raw_hw(const RawRegs& k, int d_ul, int v_u, int h_l, int t, int o,
                                            int& d, int& v, int& h) {
    const int cs = v_score(t, o, k.cmis, k.cmat);
    const int mr = d_ul + cs;
    const int xg = v_u + k.gox;
    const int yg = h_l + k.goy;
    d = v_max3(mr, xg, yg);
    v = v_max3(mr, v_u, yg);
    h = v_max3(mr, xg, h_l);
}

// carried form (nw_cell.h: cell_update_carried), compare-select score.  Plain C: hipcc selects v_max3_i32
// for the nested max itself and, unlike between asm statements, pads nothing (the predicated start-up
// groups that run this form were twice as expensive per cell as the steady ones).
__device__ __forceinline__ void cell_carried_hw(const RawRegs& k, int d_ul, int xg_u, int yg_l, int t, int o,
                                                int& d, int& xg, int& yg) {
    const int mr = d_ul + ((t == o) ? k.cmat : k.cmis);
    d = c_max3(mr, xg_u, yg_l);
    xg = max(d + k.gox, xg_u);
    yg = max(d + k.goy, yg_l);
}

__host__ __device__ inline bool fits_i8(int v) { return v >= -128 && v <= 127; }

// LDS carve of phase 1 (dynamic): code ocode[kOPad+m+kOTail] | int prog[16] | uint32 profile[waves][apad][64]
// (the rows handed from strip to strip live in the workspace -- L2 -- not in LDS: Ws2::rows)
struct P1Lds {
    size_t oc_bytes, tbl_off, total;
    __host__ __device__ explicit P1Lds(int m, int code_bytes, int tbl_bytes = 0) {
        oc_bytes = ((size_t)(kOPad + m + kOTail) * code_bytes + 15) & ~(size_t)15;
        tbl_off = oc_bytes + 64;
        total = tbl_off + (size_t)tbl_bytes;
    }
};

// ---------------------------------------------------------------------------------------------
// phase 1
//
// MODE 1: the substitution score of a cell is a compare + select on the two token ids.
// MODE 2: score PROFILE in LDS.  Each wave keeps, for the 256 rows of the strip it is in, a table
//   profile[o][lane] = the four substitution scores (signed bytes) of the lane's four rows against
//   OCR token o; a step costs one conflict-free ds_read_b32 (bank = lane) for four cells and the
//   byte is unpacked by the SDWA add that forms M^.  The table never needs rebuilding: it holds
//   "mismatch" everywhere and a strip patches the <= 4 entries per lane where its transcript tokens
//   match (and restores them when it leaves).  OCR codes sit in LDS pre-multiplied by the row pitch
//   (256 B).  Needs a small alphabet (apad x 256 B per wave) and scores that fit a signed byte.
// Both modes use the carried cell (non-positive gap opens).  A problem that does not qualify
// (a positive gap open; in MODE 2 a score outside a byte; gox != goy under SAMEGO) runs every group
// through the predicated edge body with the general cell: correct, slower, and rare.
// OC (MODE 1) = uint8_t when every token id of the batch is below 255: the OCR codes then take 1 byte
// each in LDS and a 4096-column problem fits four workgroups per CU instead of three.
template <int W, int MODE, bool SAMEGO, typename OC>
__global__ __launch_bounds__(W * 64, MODE == 1 ? 5 : 2) void nw_score_kernel(NwArgs a) {
    constexpr int R = 4;
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    constexpr bool PROFILE = (MODE == 2);
    using LC = typename std::conditional<PROFILE, uint16_t, OC>::type;      // code type in LDS
#ifndef TA_P1_CHK
#define TA_P1_CHK 4
#endif
#ifndef TA_P1_ABLATE
#define TA_P1_ABLATE 0      // timing experiments only (tools/p1_ablate.sh): any bit set breaks the results
#endif
    constexpr int ABL = TA_P1_ABLATE;
    // groups between two looks at the progress word of the strip above: a strip follows the one
    // above at CHK + 17 groups, and the ramp of a workgroup (wave w idles w x that lag at the start,
    // the waves above idle as long at the end) is what a finer grain buys back
    constexpr int CHK = TA_P1_CHK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int p = blockIdx.x;
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = (int)(a.t_off[p + 1] - t0);
    const int m = (int)(a.o_off[p + 1] - o0);
    if (n <= 0 || m <= 0) return;

    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    RawRegs kr;
    kr.cmis = prm[1] - prm[4] - prm[5]; kr.cmat = prm[0] - prm[4] - prm[5];
    kr.gox = prm[2]; kr.goy = prm[3];
    // state of a problem is kept in carried form (XG, YG) iff its gap opens are non-positive; phase 2
    // applies the same predicate when it reads checkpoints and bottom rows
    const bool carried = opens_nonpositive(kr.gox, kr.goy);
    const int xadj = carried ? kr.gox : 0, yadj = carried ? kr.goy : 0;
    const int apad = PROFILE ? a.apad : 0;
    const bool steady_ok = carried && (!PROFILE || (fits_i8(kr.cmis) && fits_i8(kr.cmat) && apad >= 2)) &&
                           (!SAMEGO || kr.gox == kr.goy);
    // keep the two select constants resident in VGPRs (hipcc otherwise re-materialises them with
    // two v_mov per step, 8 % of the loop's VALU instructions)
    if (!PROFILE) asm volatile("" : "+v"(kr.cmis), "+v"(kr.cmat));

    const P1Lds lds(m, (int)sizeof(LC));
    LC* ocode = reinterpret_cast<LC*>(smem);
    int* prog = reinterpret_cast<int*>(smem + lds.oc_bytes);
    uint32_t* tbl = reinterpret_cast<uint32_t*>(smem + lds.tbl_off);
    const Ws2 ws(n, m);
    uint8_t* const ws_p = a.ws + a.ws_off[p];
    // Bottom rows (Ws2::rows): strip s reads row s and writes row s + 1.  All waves of a workgroup
    // share one CU and its L1, so the workgroup-scope release / acquire on the progress words (LDS)
    // is all the ordering these plain loads and stores need.
    int2* const row0p = reinterpret_cast<int2*>(ws_p + ws.row(0)) + 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // codes as the cells compare them: MODE 2 keeps them multiplied by the profile's row pitch
    const int code_shift = PROFILE ? 8 : 0;
    const LC pad_code = PROFILE ? (LC)((apad - 1) << 8) : (LC)~(LC)0;        // never a valid id
    for (int j = tid; j < kOPad + m + kOTail; j += W * 64) {
        const int src = j - kOPad;
        ocode[j] = (src >= 0 && src < m) ? (LC)(a.o_codes[o0 + src] << code_shift) : pad_code;
    }
    for (int j = tid; j <= m; j += W * 64)
        row0p[j] = make_int2(raw_of(bnd_V_row0(c, j)) + xadj, raw_of(bnd_D_row0(c, j)));
    if (tid < 16) prog[tid] = 0;
    const uint32_t mis4 = (uint32_t)(kr.cmis & 0xFF) * 0x01010101u;
    if (PROFILE)                                   // (the pad code's row -- columns j <= 0 and j > m -- scores -128: see from_zero)
        for (int k = tid; k < W * apad * 64; k += W * 64) tbl[k] = ((k / 64) % apad == apad - 1) ? 0x80808080u : mis4;
    __syncthreads();

    uint32_t* const tblw = tbl + (size_t)wave * apad * 64;                   // this wave's profile
    const unsigned char* const tbl_lane = reinterpret_cast<const unsigned char*>(tblw) + lane * 4;

    // buffer descriptor of the problem's workspace, for the straight-line stores of the steady state
    // (inputs through readfirstlane: hipcc must be able to PROVE them wave-uniform, or it wraps every
    // buffer operation in a waterfall loop)
    const uint64_t wsa = reinterpret_cast<uint64_t>(ws_p);
    unsigned char* const ws_u = reinterpret_cast<unsigned char*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wsa >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wsa));
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(ws_u, 0, kWsRange, 0x00020000);
    const int nstrips = ws.nstrips, ngroups = ws.ngroups;
    const int prev_wave = (wave + W - 1) % W;
    const int g_lo = (63 + SPG - 1) / SPG;
    // The steady loop runs from the group in which the last lane has started to the END of the strip:
    // a lane that has passed its last column goes on over pad columns (the profile's all-mismatch
    // entry), and what it computes there only ever reaches lanes that are past their last column
    // too (same column, one step later), bottom-row entries beyond column m (room for 8; never read
    // for a column > m) and checkpoint words of finished lanes (phase 2 masks them).  Only the 16
    // start-up groups need the predicated edge body (0.35 ms of the 0.7 ms the edges took).
    const int g_hi = (steady_ok && ws.total < kWsRange) ? ngroups : 0;
    // The 16 start-up groups of a strip, in which lanes start one step apart, through the STEADY body as well (MODE 2):
    // a lane that has not reached column 1 yet runs over virtual columns j <= 0, and what it computes there must leave
    // its column-0 boundary state in place -- D = H = b_i = -(1 + gex) i in the carried form (YG = H~ + goy = M^ at
    // column 0), dsave from the lane above.  It does: the pad code scores -128, so M^ = b_(i-1) - 128 <= b_i; the XG chain
    // of a virtual column starts from V = -2^28 (set below) and carries max(b_i' + gox) over rows i' < i, which is <= b_i
    // because b falls with i iff gex <= -1 (the condition); so D' = max3(M^, XG, YG) = YG = b_i, YG' = max(b_i + goy,
    // b_i) = b_i, and the D handed down by the DPP shift is the b of the lane above.  The bottom-row stores of lanes 31 /
    // 63 for columns j <= 0 land in the pad entries (beyond column m + 8) of the row before: never read.  Saves the
    // EXEC-predicated edge body (twice the cost per group) on 16 of a strip's groups: 1.5 % at 4096 columns, 3 % at 2048.
#ifdef TA_P1_NO_FROM_ZERO
    const bool from_zero = false;
#else
    const bool from_zero = PROFILE && g_hi > 0 && prm[4] <= -1;
#endif
    int pass = 0;

    for (int s = wave; s < nstrips; s += W, ++pass) {
        int D[R], V[R], H[R], tc[R];            // V, H hold XG, YG when the problem is carried
        const int row0 = s * L::SR + lane * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r + 1;
            D[r] = raw_of(bnd_D_col0(c, i));
            H[r] = raw_of(bnd_H_col0(c, i)) + yadj;
            V[r] = from_zero ? -(1 << 28) : 0;
            tc[r] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
        }
        int dsave = raw_of(bnd_D_col0(c, row0));
        // retire the transcript-code loads here: otherwise hipcc parks their s_waitcnt vmcnt(0) at
        // the first use INSIDE the group loop, where it also drains every checkpoint / row store
        // of the previous group
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" :: "v"(tc[r]));
        // MODE 2: patch the profile with this strip's matches
        auto patch = [&](bool set) {
            if (!PROFILE || !steady_ok) return;
            const uint32_t matb = (uint32_t)(kr.cmat & 0xFF);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (tc[r] < 0 || tc[r] >= apad - 1) continue;
                uint32_t w = mis4;
                if (set) {
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        if (tc[q] == tc[r]) w = (w & ~(0xFFu << (8 * q))) | (matb << (8 * q));
                }
                tblw[tc[r] * 64 + lane] = w;
            }
        };
        patch(true);
        int tcmp[R];                             // transcript codes in the form the LDS codes have
#pragma unroll
        for (int r = 0; r < R; ++r) tcmp[r] = tc[r] << code_shift;           // -1 stays negative
        const bool lane_has_rows = row0 < n;
        const int prod_pass = (wave == 0) ? pass - 1 : pass;
        const int2* const hvd = reinterpret_cast<const int2*>(ws_p + ws.top(s)) + 1;       // the row above: entry j
        // the bottom row this lane writes if it is the last of its 16 (lanes 15, 31, 47: sub-strip rows for
        // phase 2; lane 63: the strip's own, which the next strip starts from)
        const bool sub_last = (lane & (kSubLanes - 1)) == kSubLanes - 1;
        int2* const hvo = reinterpret_cast<int2*>(ws_p + ws.row(s * kSubRows + (lane / kSubLanes) + 1)) + 1;

        auto wait_span = [&](int g_first) {
            if (W == 1 || s == 0) return;
            const int k_last = min((g_first + CHK + 1) * SPG - 1, L::nsteps(m) - 1);
            const int col = min(k_last + 1, m);
            const int need_groups = min(ngroups, (col + 62) / SPG + 1);
            const int need = prod_pass * ngroups + need_groups;
            while (true) {
                const int have = __hip_atomic_load(&prog[prev_wave], __ATOMIC_ACQUIRE,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(have) >= need) break;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        // Progress words say "the bottom-row entries of these groups are in L2": the stores must have
        // COMPLETED, not merely been issued, before the word is written (a workgroup-scope release
        // only orders the LDS side on this target: it emits no vmcnt wait).
        auto publish = [&](int g) {
            if (W == 1) return;
            if ((g % CHK) == 0 || g == ngroups - 1) {     // published values are 1 mod CHK, like the needs
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 63)
                    __hip_atomic_store(&prog[wave], pass * ngroups + g + 1, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        int oc_next[SPG];
        int2 hd_next[SPG];
        auto load_group = [&](int g) {
            const int idx = kOPad + g * SPG - lane;
#pragma unroll
            for (int q = 0; q < SPG; ++q) {
                oc_next[q] = ocode[idx + q];
                hd_next[q] = hvd[min(g * SPG + q + 1, m)];
            }
        };
        auto prefetch = [&](int g) {
            if (g + 1 < ngroups) {
                if (((g + 1) % CHK) == 0) wait_span(g + 1);
                load_group(g + 1);
            }
        };
        // whole lane state, taken BEFORE group g runs: what phase 2 restarts from
        auto checkpoint_now = [&](int g) {
            int* st = reinterpret_cast<int*>(ws_p + ws.state(s, g / kCkGroups)) + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) { st[r * 64] = D[r]; st[(R + r) * 64] = H[r]; }
            st[8 * 64] = V[R - 1];
            st[9 * 64] = dsave;
        };
        auto checkpoint = [&](int g) {
            if (g > 0 && (g % kCkGroups) == 0) checkpoint_now(g);
        };
        auto group_edge = [&](int g) {
            if (ABL & 32) { publish(g); return; }          // timing only: what the edge groups cost
            checkpoint(g);
            int oc[SPG];
            int2 hd[SPG];
#pragma unroll
            for (int q = 0; q < SPG; ++q) { oc[q] = oc_next[q]; hd[q] = hd_next[q]; }
            prefetch(g);
#pragma unroll
            for (int q = 0; q < SPG; ++q) {
                const int k = g * SPG + q;
                const int j = k - lane + 1;
                const bool active = (j >= 1) && (j <= m) && lane_has_rows;
                int v_up = hd[q].x, d_next = hd[q].y;
                wave_shr1_pair_sched(v_up, V[R - 1], d_next, D[R - 1]);
                if (active) {
                    int d_ul = dsave, v_u = v_up;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        if (carried) cell_carried_hw(kr, d_ul, v_u, H[r], tcmp[r], oc[q], D[r], V[r], H[r]);
                        else cell_raw_hw(kr, d_ul, v_u, H[r], tcmp[r], oc[q], D[r], V[r], H[r]);
                        d_ul = d_old;
                        v_u = V[r];
                    }
                    dsave = d_next;
                    if (sub_last) hvo[j] = make_int2(V[R - 1], D[R - 1]);
                }
            }
            publish(g);
        };

        wait_span(0);
        load_group(0);
        int g = 0;
        const int e1 = from_zero ? 0 : min(g_lo, ngroups);
        for (; g < e1; ++g) group_edge(g);

        const int g_end = g_hi & ~3;                  // steady groups run in blocks of CHK = 4 (g_lo = 16)
        if (g < g_end) {
            // ---- steady state: every lane is inside 1 <= j <= m, no EXEC changes.  Blocks of four
            // groups with two input buffers (A / B): the LDS prefetch of the next group lands in the
            // other buffer, no register copies at the back-edge, and everything that depends on the
            // group index -- progress wait and publish (once per block), checkpoint (every fourth
            // block), row / code addresses (running pointers) -- stays out of the groups.
            // MODE 2 reads the OCR codes one group further ahead (X / Y) and turns them into
            // profile entries when the group's other inputs are fetched. ----
            static_assert(CHK == 4 && SPG == 4, "the block loop is written for 4 groups of 4 steps");
            const int4* hrp = reinterpret_cast<const int4*>(hvd + (g * SPG + 1));   // next group's 4 entries (16-B aligned)
            // What lane 63 leaves behind per group -- four entries of the bottom row, two 16-byte
            // stores -- goes through a raw buffer descriptor of the problem's workspace with a per-lane
            // offset that is out of range for every other lane: the hardware drops those lanes'
            // stores, and the code stays STRAIGHT-LINE.  Under `if (lane == 63)` (a branch) hipcc cannot
            // count the stores, so its s_waitcnt for the next group's loads came out as vmcnt(2) and
            // made every group wait for its stores to be acknowledged.
            // The whole offset sits in the per-lane VGPR (+ an immediate per group of the block): with an
            // SGPR offset hipcc does not cover the store-data hazard of 16-byte buffer stores on this
            // chip (a VALU write to a data register in the two issue slots behind the store is what
            // gets stored; seen as ~12 % wrong first words when a second workgroup shares the CU).
            // (32-bit UNSIGNED arithmetic: kWsRange + any real offset < 2^32 wraps nowhere, and the sum is
            // >= kWsRange, i.e. out of the descriptor's range, for every lane but 63; g_hi != 0 only when
            // ws.total < kWsRange, so lane 63's own offsets are all in range)
            static_assert((uint64_t)kWsRange * 2 < (1ull << 32), "only63 + offset stays below 2^32");
            const uint32_t only63 = sub_last ? 0u : (uint32_t)kWsRange;           // in range for lanes 15, 31, 47, 63 only
            // entry j = k - lane + 1 of the lane's own bottom row (index j + 1)
            uint32_t vo_w = only63 + (uint32_t)(ws.row(s * kSubRows + (lane / kSubLanes) + 1) + 8 + (int64_t)(g * SPG - lane + 1) * 8);
            int crd = (kOPad + g * SPG - lane) * (int)sizeof(LC); // byte offset of the next group's codes (per lane)
            asm volatile("" : "+v"(crd));                         // a running VGPR pointer, immediate offsets below
            const unsigned char* const oc_b = reinterpret_cast<const unsigned char*>(ocode);
            const int need_cap = min(ngroups, (m + 62) / SPG + 1);
            const int need_base = prod_pass * ngroups;
            int inA[SPG], inB[SPG];              // MODE 1: OCR codes; MODE 2: packed scores of the 4 rows
            int2 hdA[SPG], hdB[SPG];
            int ocX[SPG], ocY[SPG];
            auto codes = [&](int (&oc)[SPG]) {                    // codes of the next group not yet read
#pragma unroll
                for (int q = 0; q < SPG; ++q) oc[q] = *reinterpret_cast<const LC*>(oc_b + crd + q * (int)sizeof(LC));
                crd += SPG * (int)sizeof(LC);
            };
            auto fetch = [&](const int (&oc)[SPG], int (&in)[SPG], int2 (&hd)[SPG]) {   // inputs of the next group
#pragma unroll
                for (int q = 0; q < SPG; ++q) {
                    if constexpr (PROFILE) in[q] = (ABL & 2) ? oc[q] : *reinterpret_cast<const int*>(tbl_lane + oc[q]);
                    else in[q] = oc[q];
                }
                const int4 e01 = hrp[0], e23 = hrp[1];
                hd[0] = make_int2(e01.x, e01.y); hd[1] = make_int2(e01.z, e01.w);
                hd[2] = make_int2(e23.x, e23.y); hd[3] = make_int2(e23.z, e23.w);
                hrp += 2;
            };
            auto wait_block = [&](int g_first) {                  // wait_span without the edge clamps
                if (W == 1 || s == 0 || (ABL & 8)) return;
                const int need = need_base + min(need_cap, g_first + CHK + 17);
                while (true) {
                    const int have = __hip_atomic_load(&prog[prev_wave], __ATOMIC_ACQUIRE,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__builtin_amdgcn_readfirstlane(have) >= need) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            };
            auto steady = [&](auto blk, const int (&in)[SPG], const int2 (&hd)[SPG]) {
                constexpr int B = decltype(blk)::value;           // group's place in the block: immediate offsets
                int bv[SPG], bd[SPG];
#pragma unroll
                for (int q = 0; q < SPG; ++q) {
                    int x_up = hd[q].x, d_next = hd[q].y;
                    if (!(ABL & 16)) wave_shr1_pair_sched(x_up, V[R - 1], d_next, D[R - 1]);
                    else { x_up += V[R - 1]; d_next += D[R - 1]; }
                    int d_ul = dsave, x_u = x_up;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        int mr;
                        if constexpr (PROFILE) mr = c_add_sbyte(d_ul, in[q], r);
                        else mr = d_ul + v_score(tc[r], in[q], kr.cmis, kr.cmat);
                        const int d = c_max3(mr, x_u, H[r]);
                        const int dgx = d + kr.gox;
                        const int dgy = SAMEGO ? dgx : d + kr.goy;
                        x_u = max(dgx, x_u);
                        H[r] = max(dgy, H[r]);
                        V[r] = x_u;
                        D[r] = d;
                        d_ul = d_old;
                    }
                    dsave = d_next;
                    bv[q] = V[R - 1]; bd[q] = D[R - 1];
                }
                typedef int v4i __attribute__((ext_vector_type(4)));
                if (!(ABL & 1)) {                                  // the strip's bottom row, by its owner
                    __builtin_amdgcn_raw_buffer_store_b128((v4i){bv[0], bd[0], bv[1], bd[1]}, wsrc, (int)(vo_w + B * 32u), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128((v4i){bv[2], bd[2], bv[3], bd[3]}, wsrc, (int)(vo_w + B * 32u + 16u), 0, 0);
                }
            };
            // (hvd of group g was waited for by the last edge group's prefetch)
            if constexpr (PROFILE) { codes(ocX); codes(ocY); fetch(ocX, inA, hdA); }
            else { codes(ocX); fetch(ocX, inA, hdA); }
            constexpr std::integral_constant<int, 0> blk0{};
            constexpr std::integral_constant<int, 1> blk1{};
            constexpr std::integral_constant<int, 2> blk2{};
            constexpr std::integral_constant<int, 3> blk3{};
            auto block = [&]() {                                  // four groups, g .. g + 3
                if constexpr (PROFILE) {
                    fetch(ocY, inB, hdB); codes(ocX); steady(blk0, inA, hdA);
                    fetch(ocX, inA, hdA); codes(ocY); steady(blk1, inB, hdB);
                    fetch(ocY, inB, hdB); codes(ocX); steady(blk2, inA, hdA);
                    wait_block(g + CHK);
                    fetch(ocX, inA, hdA); codes(ocY); steady(blk3, inB, hdB);
                } else {
                    codes(ocY); fetch(ocY, inB, hdB); steady(blk0, inA, hdA);
                    codes(ocX); fetch(ocX, inA, hdA); steady(blk1, inB, hdB);
                    codes(ocY); fetch(ocY, inB, hdB); steady(blk2, inA, hdA);
                    wait_block(g + CHK);
                    codes(ocX); fetch(ocX, inA, hdA); steady(blk3, inB, hdB);
                }
                // progress, one block late: this block issued 8 stores and 8 loads (16 vector-memory
                // operations, + 10 checkpoint stores in every fourth block), so once all but the 16
                // youngest are done, every store of the blocks before this one has completed -- without
                // waiting for the stores just issued.  Published: groups < g.
                vo_w += (uint32_t)(CHK * SPG * 8);
                if (W > 1 && !(ABL & 8)) {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    if (lane == 63)
                        __hip_atomic_store(&prog[wave], pass * ngroups + g, __ATOMIC_RELEASE,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                g += CHK;
            };
            // Checkpoint intervals whole, so that the ten
            // checkpoint stores sit in straight-line code: behind a branch hipcc's s_waitcnt for the
            // next loads can no longer count them and waits for all of them (a full store round trip
            // every 16 groups).
            static_assert(kCkGroups % CHK == 0, "interval = whole blocks");
            while (g < g_end && (g % kCkGroups) != 0) block();       // up to the first interval border (g_lo = 16)
            while (g + kCkGroups <= g_end) {
                if (!(ABL & 4)) checkpoint_now(g);
#pragma unroll
                for (int b = 0; b < kCkGroups / CHK; ++b) block();
            }
            while (g < g_end) {
                if (!(ABL & 4)) checkpoint(g);
                block();
            }
            if (g < ngroups) {
                if ((g % CHK) == 0) wait_span(g);
                load_group(g);
            } else if (W > 1) {                              // the strip ended in the block loop: all of it is out
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 63)
                    __hip_atomic_store(&prog[wave], pass * ngroups + ngroups, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        for (; g < ngroups; ++g) group_edge(g);
        patch(false);
    }
}

// ---------------------------------------------------------------------------------------------
// phase 2
// One wave per problem walks back strip by strip, and inside a strip CHUNK by chunk (a chunk =
// the kCkGroups groups between two state checkpoints).  For the chunk the walk is in, the wave
// re-runs the TAGGED cell from the chunk's checkpoint -- straight into LDS: the chunk's pointer
// bytes (16 KiB) never go to memory -- walks them, and steps to the chunk before when the walk
// reaches the chunk's first two steps (the restart state carries no winner tags; two steps later
// every input of a cell has been produced with tags).  Those two steps belong to the chunk before,
// whose re-fill therefore runs one group further (17 groups).
constexpr int kChunk = kCkGroups;                 // groups per chunk
constexpr int kChunkGroups = kChunk + 1;          // + the group holding the next chunk's two halo steps
constexpr int kChunkSteps = kChunkGroups * 4;
// Lanes of a chunk whose pointer bytes are kept.  The walk enters a chunk at a lane l and only ever
// moves to smaller lanes (a step up is a quarter of a lane, a step left none), a quarter lane per
// alignment column at most: the 64 .. 68 skewed steps of a chunk take a diagonal path across ~13 lanes.
// Keeping lanes l - 31 .. l instead of all 64 halves the chunk's LDS (10.1 KB per wave with the ops
// staging buffer gone: the walk writes its columns straight to memory), i.e. SIXTEEN traceback waves per
// CU instead of eight -- the 16 problems a CU gets at 4096 problems are then all resident at once.  A walk
// that does run off the window's top lane (a long run of transcript-side gaps) re-fills the same chunk,
// up to the group it stands in, with the window moved to its lane.
#ifndef TA_TB2_LANES
#define TA_TB2_LANES 32
#endif
constexpr int kWinLanes = TA_TB2_LANES;
static_assert(kWinLanes >= 8 && kWinLanes <= 64, "window lanes");

#ifndef TA_P2_PROFILE
#define TA_P2_PROFILE 0     // cycle counters per problem into row 0 of its workspace (tools/p2_profile.py)
#endif
#ifndef TA_P2_ABLATE
#define TA_P2_ABLATE 0      // timing experiments only: 2 re-fill one group only, 4 no walk
#endif

template <bool CARRIED, bool SAMEGO, int WL = kWinLanes>
__device__ __forceinline__ void refill_chunk(const CellRegs& kr, int (&D)[4], int (&V)[4], int (&H)[4], int& dsave,
                                             const int (&tc)[4], const int2* hvt, const uint16_t* ow,
                                             uint4* win, int2* hvb, int g0, int g_top, int m, int lane,
                                             bool lane_has_rows, int l_lo, int top_steps) {
    constexpr int R = 4, SPG = 4;
    const int k0 = g0 * SPG;
    int oc_next[SPG];
    int2 hd_next[SPG];
    auto load_group = [&](int g) {
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            const int kk = g * SPG + q;
            oc_next[q] = ow[kk - k0 + 63 - lane];
            hd_next[q] = hvt[min(kk + 1, m) - k0];
        }
    };
    auto cell = [&](int d_ul, int x_u, int y_l, int t, int o, int& d, int& x, int& y) -> unsigned {
        // (the asm-pinned forms: with four waves per SIMD hipcc's own scheduling of the C forms measured 2 % slower here)
        if constexpr (CARRIED) return cell_carried_tagged_hw<SAMEGO>(kr, d_ul, x_u, y_l, t, o, d, x, y);
        else return cell_hw(kr, d_ul, x_u, y_l, t, o, d, x, y);
    };
    const bool in_win = (unsigned)(lane - l_lo) < (unsigned)WL;
    load_group(g0);
    // unpredicated groups: from the one in which the last lane has started on (a lane past its last column
    // goes on over pad codes; what it computes reaches only lanes that are past theirs, pointer bytes
    // and captured bottom-row entries of columns > m, none of which is ever read -- as in phase 1)
    const int gs_lo = (63 + SPG - 1) / SPG;
    // one unpredicated group of NQ <= 4 steps (the steps behind the walk's entry point are nobody's: the walk only
    // moves to smaller k, and the lane state is dropped after the chunk)
    auto steady_group = [&](int g, const int (&oc)[SPG], const int2 (&hd)[SPG], unsigned (&acc)[4], auto nq_c) {
        constexpr int NQ = decltype(nq_c)::value;
        int2 cap[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int v_up = hd[q].x, d_next = hd[q].y;
            wave_shr1_pair_sched(v_up, V[R - 1], d_next, D[R - 1]);
            int d_ul = dsave, v_u = v_up;
            unsigned b[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int d_old = D[rr];
                b[rr] = cell(d_ul, v_u, H[rr], tc[rr], oc[q], D[rr], V[rr], H[rr]);
                d_ul = d_old;
                v_u = V[rr];
            }
            acc[q] = pack4(b[0], b[1], b[2], b[3]);
            dsave = d_next;
            cap[q] = make_int2(V[R - 1], D[R - 1]);
        }
        if (lane == 63) {                               // its bottom-row outputs of the steps
#pragma unroll
            for (int q = 0; q < NQ; ++q) hvb[g * SPG - k0 + q] = cap[q];
        }
    };
    auto edge_group = [&](int g, const int (&oc)[SPG], const int2 (&hd)[SPG], unsigned (&acc)[4]) {
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            const int kk = g * SPG + q;
            const int j = kk - lane + 1;
            const bool active = (j >= 1) && (j <= m) && lane_has_rows;
            int v_up = hd[q].x, d_next = hd[q].y;
            wave_shr1_pair<4>(v_up, V[R - 1], d_next, D[R - 1]);
            if (active) {
                int d_ul = dsave, v_u = v_up;
                unsigned b[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int d_old = D[rr];
                    b[rr] = cell(d_ul, v_u, H[rr], tc[rr], oc[q], D[rr], V[rr], H[rr]);
                    d_ul = d_old;
                    v_u = V[rr];
                }
                acc[q] = pack4(b[0], b[1], b[2], b[3]);
                dsave = d_next;
                if (lane == 63) hvb[kk - k0] = make_int2(V[R - 1], D[R - 1]);
            }
        }
    };
    for (int g = g0; g <= g_top; ++g) {
        int oc[SPG];
        int2 hd[SPG];
#pragma unroll
        for (int q = 0; q < SPG; ++q) { oc[q] = oc_next[q]; hd[q] = hd_next[q]; }
        unsigned acc[4] = {0u, 0u, 0u, 0u};
        if (g < g_top) {
            load_group(g + 1);
            if (g >= gs_lo) steady_group(g, oc, hd, acc, std::integral_constant<int, SPG>{});
            else edge_group(g, oc, hd, acc);
        } else if (g >= gs_lo) {
            // the group the walk enters the chunk in: only its steps up to the entry point (top_steps of them)
            if (top_steps <= 2) steady_group(g, oc, hd, acc, std::integral_constant<int, 2>{});
            else steady_group(g, oc, hd, acc, std::integral_constant<int, SPG>{});
        } else {
            edge_group(g, oc, hd, acc);
        }
        if (WL == 64 || in_win) win[(g - g0) * WL + (lane - l_lo)] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    }
}

__global__ __launch_bounds__(64) void nw_trace2_kernel(NwArgs a) {
    constexpr int ABL2 = TA_P2_ABLATE;
    constexpr int R = 4;
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    __shared__ uint4 win[kChunkGroups * kWinLanes];             // pointer bytes of the chunk: [group][lane - l_lo]
    __shared__ int2 hvt[kChunkSteps + 8];                       // (V~ or XG, D) of the row above, columns k0..
    __shared__ int2 hvb[kChunkSteps];                           // tagged (V~ or XG, D) the strip's bottom row puts out, per step
    __shared__ uint16_t ow[kChunkSteps + 64 + 8];               // OCR codes, o index (k0 - 63) + i

    const int p = blockIdx.x, lane = threadIdx.x;
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = (int)(a.t_off[p + 1] - t0);
    const int m = (int)(a.o_off[p + 1] - o0);
    uint8_t* ops = a.ops_out + a.ops_off[p];
    const int cap = n + m;
    int x = n, y = m, len = 0, st = 0;
    bool first = true;
    // The bottom rows of phase 1 carry no winner tags, so the two pointers a strip's first row takes
    // from the row above (PM, PX) are not in a re-filled window.  A step that leaves the strip
    // upwards is taken with its next state PENDING (pend = 3: the tag of D, = 4: the tag of XG / V~,
    // of the cell (x, y) the walk then stands on -- bottom row of the strip above), and the state is
    // read off that cell's tagged outputs (hvb) when the strip above is re-filled.
    int pend = 0;
    bool probe = false;
    if (n > 1 && (n - 1) % L::SR == 0) {                        // the start state PM(n, m) is such a tag: D(n-1, m-1)
        if (m == 1) { st = 0; first = false; }                  // boundary column: M
        else { x = n - 1; y = m - 1; pend = 3; probe = true; }
    }

    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    CellRegs kr;
    kr.cmis = c.cmismatch; kr.cmat = c.cmatch; kr.gox6 = c.gox6; kr.goy6 = c.goy6;
    kr.clean = ~kTagMask;
    // phase 1 leaves V~ + gox / H~ + goy in its checkpoints and bottom rows when the gap opens are
    // non-positive (carried cell), and the chunks are then re-filled in that form too
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int xadj = carried ? c.gox : 0, yadj = carried ? c.goy : 0;
    const int xadj6 = xadj * 64, yadj6 = yadj * 64;
    const Ws2 ws(max(n, 1), max(m, 1));
    uint8_t* const ws_p = a.ws + a.ws_off[p];
#if TA_P2_PROFILE
    long long pc_setup = 0, pc_fill = 0, pc_walk = 0, pc_chunks = 0, pc_t = __builtin_readcyclecounter(), pc_groups = 0;
    const long long pc_start = pc_t;
    long long pc_iters = 0;
#define PC_LAP(acc) { const long long now_ = __builtin_readcyclecounter(); acc += now_ - pc_t; pc_t = now_; }
#else
#define PC_LAP(acc)
#endif

    while (x > 0 && y > 0) {
        const int s = (x - 1) / L::SR;
        int l = ((x - 1) % L::SR) / R;
        int r = (x - 1) % R;
        int k = (y - 1) + l;
        const int i_h = s * L::SR;                              // 1-based index of the row above the strip
        const int row0 = s * L::SR + lane * R;
        const bool lane_has_rows = row0 < n;
        int tc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = row0 + rr + 1;
            tc[rr] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
        }
        const int2* const hrow = reinterpret_cast<const int2*>(ws_p + ws.top(s)) + 1;     // the row above: entry j

        int ck = (k >> 2) / kChunk;                             // chunk the walk is in
        int g_top = k >> 2;                                     // last group to re-fill
        bool in_strip = true;
        // Inputs of a chunk's re-fill, fetched into registers: the OCR codes ow[i] = o[(k0 - 63) + i], the
        // row above the strip for columns k0 .. min(m, k0 + steps) (tagged only where the tags are known
        // analytically: the table's boundary row) and the lane state at the chunk's first group.  The
        // chunk after this one is nearly always the one before it in the same strip, whole: its inputs
        // are requested as soon as this chunk's are in LDS and arrive under this chunk's re-fill and walk.
        constexpr int kOwIt = (kChunkSteps + 64 + 63) / 64, kRowIt = (kChunkSteps + 1 + 63) / 64;
        int in_ow[kOwIt], in_st[kStateInts];
        int2 in_row[kRowIt];
        int in_ck = -1, in_gtop = -1;                           // what the registers hold (this strip)
        auto fetch_inputs = [&](int ck_, int gtop_) {
            const int g0_ = ck_ * kChunk, k0_ = g0_ * SPG;
            const int nsteps_ = (gtop_ - g0_ + 1) * SPG;
#pragma unroll
            for (int it = 0; it < kOwIt; ++it) {
                const int src = k0_ - 63 + it * 64 + lane;
                in_ow[it] = (src >= 0 && src < m) ? a.o_codes[o0 + src] : 0xFFFF;
            }
            const int jhi = min(m, k0_ + nsteps_);
#pragma unroll
            for (int it = 0; it < kRowIt; ++it) {
                const int j = min(k0_ + it * 64 + lane, jhi);
                if (s == 0) in_row[it] = make_int2(bnd_V_row0(c, j) + xadj6, bnd_D_row0(c, j));
                else {
                    const int2 e = hrow[max(j, 1)];
                    in_row[it] = (j == 0) ? make_int2(0, bnd_D_col0(c, i_h)) : make_int2(enc_of(e.x), enc_of(e.y));
                }
            }
            if (g0_ > 0) {
                const int* stp = reinterpret_cast<const int*>(ws_p + ws.state(s, g0_ / kCkGroups)) + lane;
#pragma unroll
                for (int q = 0; q < kStateInts; ++q) in_st[q] = stp[q * 64];
            }
            in_ck = ck_; in_gtop = gtop_;
        };
        while (in_strip) {
            const int g0 = ck * kChunk;
            const int k0 = g0 * SPG;
            const int kvalid = ck > 0 ? k0 + 2 : 0;
            const int nsteps_w = (g_top - g0 + 1) * SPG;
            const int l_lo = kWinLanes == 64 ? 0 : max(0, l - (kWinLanes - 1));   // the window: lanes l_lo .. l_lo + kWinLanes - 1
            if (in_ck != ck || in_gtop != g_top) fetch_inputs(ck, g_top);

            // (a) OCR codes of the chunk, (b) the row above the strip
#pragma unroll
            for (int it = 0; it < kOwIt; ++it) {
                const int i = it * 64 + lane;
                if (i < nsteps_w + 64) ow[i] = (uint16_t)in_ow[it];
            }
#pragma unroll
            for (int it = 0; it < kRowIt; ++it) {
                const int j = k0 + it * 64 + lane;
                if (j <= min(m, k0 + nsteps_w)) hvt[j - k0] = in_row[it];
            }
            // (c) lane state at the start of group g0 (in the form the re-fill keeps: carried or not)
            int D[R], V[R], H[R];
            int dsave;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int i = row0 + rr + 1;
                V[rr] = 0;
                D[rr] = bnd_D_col0(c, i);
                H[rr] = bnd_H_col0(c, i) + yadj6;
            }
            dsave = bnd_D_col0(c, row0);
            // a lane that has not started by the chunk's first step (lane >= k0: first chunks of a strip
            // when the interval is shorter than 64 steps) keeps the boundary values above: the scores are
            // the same, and only this form carries the column-0 tags the lane's first cells point to
            static_assert(kStateInts == 2 * R + 2, "state = D[R], H[R], V[R-1], dsave");
            if (g0 > 0 && lane < k0) {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) { D[rr] = enc_of(in_st[rr]); H[rr] = enc_of(in_st[R + rr]); }
                V[R - 1] = enc_of(in_st[2 * R]);
                dsave = enc_of(in_st[2 * R + 1]);
            }
            if (ck > 0) fetch_inputs(ck - 1, g0);              // the likely next chunk (registers are free again)
            __syncthreads();
            PC_LAP(pc_setup)

            // (d) tagged re-fill of groups g0 .. g_top into LDS; of group g_top only the steps up to the walk's (k & 3)
            {
                const int top_steps = ((k >> 2) == g_top) ? (k & 3) + 1 : SPG;
                if (carried && c.gox == c.goy) refill_chunk<true, true>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, l_lo, top_steps);
                else if (carried) refill_chunk<true, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, l_lo, top_steps);
                else refill_chunk<false, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, l_lo, top_steps);
            }
            __syncthreads();
            PC_LAP(pc_fill)
#if TA_P2_PROFILE
            pc_chunks += 1; pc_groups += g_top - g0 + 1;
#endif

            // (e) walk the chunk
            const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
            if (pend) {                                        // (x, y): lane 63's last row, step k of this chunk
                const int2 e = hvb[k - k0];
                st = 2 - (((pend == 3) ? e.y : e.x) & 3);
                pend = 0;
                if (probe) {                                   // that was the start state: back to (n, m)
                    probe = false; first = false;
                    x = n; y = m;
                    __syncthreads();
                    break;
                }
            }
            if (first && k >= kvalid) {                        // start state, textSeqCompare.py:102
                st = ptr_pm(wb[(((k >> 2) - g0) * kWinLanes + (l - l_lo)) * 16 + (k & 3) * R + r]);
                first = false;
            }
            if (ABL2 & 4) { x = s * L::SR; y = max(y - 300, 1); }
            if (!(ABL2 & 4)) {                                  // columns go straight to the right-aligned output
#if TA_P2_PROFILE
                long long* const itp = &pc_iters;
#else
                long long* const itp = nullptr;
#endif
                len += walk_window_vec<true, kWinLanes, true>(win, g0, kvalid, s * L::SR, x, y, st,
                                                              ops + (cap - 1 - len), cap - len, lane, itp, l_lo);
            }
            if (st >= 3) { pend = st; st = 0; }                // left the strip upwards: state pending
            PC_LAP(pc_walk)
            // position in layout coordinates after the walk
            l = (x > s * L::SR) ? ((x - 1) % L::SR) / R : -1;
            r = (x - 1) & (R - 1);
            k = (y - 1) + l;
            if ((x <= 0) | (y <= 0) | (l < 0)) {
                in_strip = false;                              // the walk left the strip or finished
            } else if (k >= kvalid) {
                // still inside this chunk: the walk ran off the top lane of the window.  The same chunk
                // again, up to the group the walk stands in, with the window at its lane.
                g_top = k >> 2;
            } else {
                // it ran into the chunk's two halo steps (k < kvalid): the chunk before, extended by
                // the group that holds them
                g_top = g0;
                ck -= 1;
            }
        }
    }
    while (y > 0) { if (lane == 0) ops[cap - 1 - len] = 2; ++len; --y; }
    while (x > 0) { if (lane == 0) ops[cap - 1 - len] = 1; ++len; --x; }
    if (lane == 0) a.ops_len[p] = len;
#if TA_P2_PROFILE
    if (lane == 0) {                                   // row 0 of the workspace is phase 1's: free by now
        long long* out = reinterpret_cast<long long*>(ws_p + ws.row(0));
        out[0] = pc_setup; out[1] = pc_fill; out[2] = pc_walk; out[3] = pc_chunks; out[4] = pc_groups; out[5] = len;
        out[6] = pc_iters; out[7] = __builtin_readcyclecounter() - pc_start;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// phase 2 for SMALL and MEDIUM batches: NWV waves per problem, speculating along the path.
//
// With one wave per problem (above) a problem's traceback is a chain of ~6 chunks per strip, each
// set-up -> re-fill (70 %) -> walk, and at a thousand problems that lone wave per SIMD IS the launch's
// time (1024 x 2048^2: 0.70 ms, 64 problems: 0.63 ms -- latency, not throughput).  But which chunk
// comes next is almost always known before the walk gets there: the chunk BEFORE the one it is in,
// same strip, entered through the two halo steps.  So the chunks along the path are dealt to the
// waves of a workgroup round-robin -- iteration i (the i-th chunk of the path) belongs to wave
// i mod NWV.  A wave re-fills the chunk it EXPECTS its iteration to be (the newest job it knows, moved
// back by the iterations in between) into its own LDS window while the waves before it are still
// busy, then waits for the token of iteration i - 1 (position, state, alignment length so far and
// the job of iteration i as the walk really left it), re-fills again only if the expectation was
// wrong (the walk left the strip: once per strip), walks, and passes the token on.  Speculative
// re-fills keep all 64 lanes of a chunk (the entry lane is not known yet; 17 KiB of LDS per wave --
// what a batch this small can afford) and, of the halo group, the two steps a walk can enter at.
// Results are the one-wave kernel's, bit for bit: same re-fill, same walk, same pending-state rules.
struct TbJob { int s, ck, gtop, tops; };                     // strip, chunk, last group to re-fill, steps of that group (1..4); s < 0: none
__device__ __forceinline__ TbJob tb_prev_job(const TbJob& j) {
    // the chunk before, entered through its halo: groups up to the first group of the chunk just left, two steps of it
    if (j.s < 0 || j.ck < 1) return TbJob{-1, 0, 0, 0};
    return TbJob{j.s, j.ck - 1, j.ck * kChunk, 2};
}
__device__ __forceinline__ TbJob tb_job_at(int x, int y) {
    constexpr int R = 4, SR = 256;
    if (x <= 0 || y <= 0) return TbJob{-1, 0, 0, 0};
    const int s = (x - 1) / SR, l = ((x - 1) % SR) / R, k = (y - 1) + l;
    int ck = (k >> 2) / kChunk;
    // the first two steps of a chunk carry no valid tags: they belong to the chunk before, whose re-fill runs
    // one group further (the one-wave kernel finds that out by a walk of zero steps)
    if (ck > 0 && k < ck * kChunk * 4 + 2) ck -= 1;
    return TbJob{s, ck, k >> 2, (k & 3) + 1};
}

// The job d iterations after the one that starts at (x, y) as job j, if the path keeps to the diagonal: chunk after
// chunk through the halo while the strip lasts (k falls by 5/4 per diagonal step: a column and a quarter lane), then
// the chunk of the strip above that the diagonal enters -- with two groups of margin on its entry group, whole groups:
// a re-fill that went further than the walk's entry point serves it as well (tb_serves).
__device__ __forceinline__ TbJob tb_predict(int x, int y, TbJob j, int d) {
    for (; d > 0 && j.s >= 0; --d) {
        const int l = ((x - 1) % 256) / 4, k = (y - 1) + l;
        const int to_top = x - j.s * 256;                        // diagonal steps until the walk leaves the strip
        const int to_halo = j.ck > 0 ? ((k - (j.ck * kChunk * 4 + 1)) * 4 + 4) / 5 : (1 << 28);
        if (to_halo < to_top) {
            x -= to_halo; y -= to_halo;
            j = (y > 0) ? tb_prev_job(j) : TbJob{-1, 0, 0, 0};
        } else {
            x -= to_top; y -= to_top;
            j = tb_job_at(x, y);
            if (j.s >= 0) { j.gtop = min(j.gtop + 2, j.ck * kChunk + kChunk); j.tops = 4; }
        }
    }
    return j;
}
// a window re-filled for `spec` holds everything the walk of job `j` reads: same chunk, re-filled at least as far
__device__ __forceinline__ bool tb_serves(const TbJob& spec, const TbJob& j) {
    return spec.s >= 0 && spec.s == j.s && spec.ck == j.ck &&
           (spec.gtop > j.gtop || (spec.gtop == j.gtop && spec.tops >= j.tops));
}

struct TbTok { int x, y, st, len, pend, flags; TbJob job; };  // flags: 1 first, 2 probe, 4 done
constexpr int kTokInts = 10;

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void nw_trace2w_kernel(NwArgs a) {
    constexpr int R = 4;
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    constexpr int WL = 64;                                      // whole chunks: the entry lane of a speculative chunk is unknown
    __shared__ uint4 win_s[NWV][kChunkGroups * WL];
    __shared__ int2 hvt_s[NWV][kChunkSteps + 8];
    __shared__ int2 hvb_s[NWV][kChunkSteps];
    __shared__ uint16_t ow_s[NWV][kChunkSteps + 64 + 8];
    __shared__ int tok_s[NWV][kTokInts];
    __shared__ int seq_s;                                       // iterations whose token is published

    const int p = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = (int)(a.t_off[p + 1] - t0);
    const int m = (int)(a.o_off[p + 1] - o0);
    uint8_t* ops = a.ops_out + a.ops_off[p];
    const int cap = n + m;
    if (threadIdx.x == 0) seq_s = 0;
    __syncthreads();                                            // the only workgroup barrier: the waves run apart from here

    uint4* const win = win_s[wave];
    int2* const hvt = hvt_s[wave];
    int2* const hvb = hvb_s[wave];
    uint16_t* const ow = ow_s[wave];
    // one wave's LDS writes followed by its own lanes' reads: the LDS executes a wave's operations in order; this
    // only keeps the compiler from moving them across
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    CellRegs kr;
    kr.cmis = c.cmismatch; kr.cmat = c.cmatch; kr.gox6 = c.gox6; kr.goy6 = c.goy6;
    kr.clean = ~kTagMask;
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int xadj = carried ? c.gox : 0, yadj = carried ? c.goy : 0;
    const int xadj6 = xadj * 64, yadj6 = yadj * 64;
    const Ws2 ws(max(n, 1), max(m, 1));
    uint8_t* const ws_p = a.ws + a.ws_off[p];

    // the token before iteration 0: the start of the walk (textSeqCompare.py:100-107), the same on every wave
    TbTok T;
    T.x = n; T.y = m; T.st = 0; T.len = 0; T.pend = 0; T.flags = 1;
    if (n > 1 && (n - 1) % L::SR == 0) {                        // the start state PM(n, m) is a tag of the strip above
        if (m == 1) T.flags = 0;                                // boundary column: M
        else { T.x = n - 1; T.y = m - 1; T.pend = 3; T.flags = 1 | 2; }
    }
    T.job = tb_job_at(T.x, T.y);
    if (T.job.s < 0) {                                          // an empty string: nothing to walk, boundary run only
        if (wave == 0) {
            int x = n, y = m, len = 0;
            while (y > 0) { if (lane == 0) ops[cap - 1 - len] = 2; ++len; --y; }
            while (x > 0) { if (lane == 0) ops[cap - 1 - len] = 1; ++len; --x; }
            if (lane == 0) a.ops_len[p] = len;
        }
        return;
    }
    int kt = -1;                                                // index of the newest token this wave holds

    // re-fill of one job into this wave's window (set-up + tagged fill), as the one-wave kernel does it
    auto refill = [&](const TbJob& J) {
        const int s = J.s, g0 = J.ck * kChunk, k0 = g0 * SPG, g_top = J.gtop;
        const int nsteps_w = (g_top - g0 + 1) * SPG;
        const int i_h = s * L::SR;
        const int row0 = s * L::SR + lane * R;
        const bool lane_has_rows = row0 < n;
        int tc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = row0 + rr + 1;
            tc[rr] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
        }
        const int2* const hrow = reinterpret_cast<const int2*>(ws_p + ws.top(s)) + 1;
        constexpr int kOwIt = (kChunkSteps + 64 + 63) / 64, kRowIt = (kChunkSteps + 1 + 63) / 64;
#pragma unroll
        for (int it = 0; it < kOwIt; ++it) {
            const int i = it * 64 + lane, src = k0 - 63 + i;
            if (i < nsteps_w + 64) ow[i] = (uint16_t)((src >= 0 && src < m) ? a.o_codes[o0 + src] : 0xFFFF);
        }
        const int jhi = min(m, k0 + nsteps_w);
#pragma unroll
        for (int it = 0; it < kRowIt; ++it) {
            const int jj = k0 + it * 64 + lane;
            if (jj <= jhi) {
                int2 v;
                if (s == 0) v = make_int2(bnd_V_row0(c, jj) + xadj6, bnd_D_row0(c, jj));
                else {
                    const int2 e = hrow[max(jj, 1)];
                    v = (jj == 0) ? make_int2(0, bnd_D_col0(c, i_h)) : make_int2(enc_of(e.x), enc_of(e.y));
                }
                hvt[jj - k0] = v;
            }
        }
        int D[R], V[R], H[R], dsave;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = row0 + rr + 1;
            V[rr] = 0;
            D[rr] = bnd_D_col0(c, i);
            H[rr] = bnd_H_col0(c, i) + yadj6;
        }
        dsave = bnd_D_col0(c, row0);
        if (g0 > 0 && lane < k0) {
            const int* stp = reinterpret_cast<const int*>(ws_p + ws.state(s, g0 / kCkGroups)) + lane;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) { D[rr] = enc_of(stp[rr * 64]); H[rr] = enc_of(stp[(R + rr) * 64]); }
            V[R - 1] = enc_of(stp[2 * R * 64]);
            dsave = enc_of(stp[(2 * R + 1) * 64]);
        }
        wave_sync();
        if (carried && c.gox == c.goy) refill_chunk<true, true, WL>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, 0, J.tops);
        else if (carried) refill_chunk<true, false, WL>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, 0, J.tops);
        else refill_chunk<false, false, WL>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lane, lane_has_rows, 0, J.tops);
        wave_sync();
    };

    // every iteration publishes exactly one token, so no wave ever waits for one that does not come; the bound is a
    // belt against a walk that makes no progress (none known): the wave that reaches it ends the walk for everyone
    const int max_iter = 4 * (L::nstrips(max(n, 1)) * (ws.ngroups / kChunk + 2)) + 64;
    for (int i = wave; ; i += NWV) {
        // (1) speculation + (2) the token of iteration i - 1.  While that token is not there, the wave re-fills the
        // job its iteration will most likely be: the job of the NEWEST token published (iteration j < i - 1), moved
        // on by the i - 1 - j iterations in between along the diagonal (tb_predict: the chunk before while the strip
        // lasts, then the chunk of the strip above the diagonal enters).  A newer token that changes the expectation (the walk left the
        // strip) replaces the speculation at once -- the waves behind the one that must re-fill the new strip's
        // first chunk re-fill its second, third ... beside it instead of one after the other.
        TbJob spec{-1, 0, 0, 0};
        if (i > kt + 1) {
            int spins = 0, based_on = -2;                         // token the current expectation comes from (-2: none yet)
            while (true) {
                const int have = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&seq_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (have >= i) break;                            // token i - 1 is there
                const int from = max(kt, have - 1);              // newest token to go by: the one held, or a fresher one
                if (from != based_on) {
                    based_on = from;
                    TbJob want = T.job;                           // job of iteration from + 1, starting at (px, py)
                    int px = T.x, py = T.y;
                    bool over = T.flags & 4;
                    if (from != kt) {                             // (its slot is not rewritten before iteration from + NWV > i)
                        const int* tj = tok_s[from % NWV];
                        want = TbJob{tj[6], tj[7], tj[8], tj[9]};
                        px = tj[0]; py = tj[1];
                        over = tj[5] & 4;
                    }
                    want = tb_predict(px, py, want, i - from - 1);
                    if (!over && want.s >= 0 &&
                        !(spec.s >= 0 && want.s == spec.s && want.ck == spec.ck && want.gtop == spec.gtop)) {
                        spec = want;
                        refill(spec);
                        continue;
                    }
                }
                // (bounded: every iteration publishes a token, so the wait always ends -- the bound only makes sure a
                // mistake here could never leave waves spinning on the chip; ~1 s)
                if (++spins >= (1 << 24)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (spins >= (1 << 24)) break;
            const int* tk = tok_s[(i - 1) % NWV];
            T.x = tk[0]; T.y = tk[1]; T.st = tk[2]; T.len = tk[3]; T.pend = tk[4]; T.flags = tk[5];
            T.job = TbJob{tk[6], tk[7], tk[8], tk[9]};
            kt = i - 1;
        }
        if (T.flags & 4) {
            // the walk is over (another wave finished it).  Pass the word on as the token of THIS iteration: the
            // wave of iteration i + 1 is waiting for it (every iteration publishes exactly one token)
            if (lane == 0) {
                tok_s[i % NWV][5] = 4;
                __hip_atomic_store(&seq_s, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            break;
        }
        // (3) the chunk itself, unless the speculation was right
        const TbJob J = T.job;
        if (!tb_serves(spec, J)) refill(J);
        // (4) walk (nw_trace2_kernel, step (e))
        int x = T.x, y = T.y, st = T.st, len = T.len, pend = T.pend;
        bool first = T.flags & 1, probe = T.flags & 2;
        const int s = J.s, g0 = J.ck * kChunk, k0 = g0 * SPG;
        const int kvalid = J.ck > 0 ? k0 + 2 : 0;
        const int l = ((x - 1) % L::SR) / R, r = (x - 1) % R, k = (y - 1) + l;
        bool walked = true;
        if (pend) {                                              // (x, y): lane 63's last row, step k of this chunk
            const int2 e = hvb[k - k0];
            st = 2 - (((pend == 3) ? e.y : e.x) & 3);
            pend = 0;
            if (probe) { probe = false; first = false; x = n; y = m; walked = false; }    // that was the start state
        }
        if (walked) {
            if (first && k >= kvalid) {                          // start state, textSeqCompare.py:102
                st = ptr_pm(reinterpret_cast<const uint8_t*>(win)[(((k >> 2) - g0) * WL + l) * 16 + (k & 3) * R + r]);
                first = false;
            }
            len += walk_window_vec<true, WL, true>(win, g0, kvalid, s * L::SR, x, y, st, ops + (cap - 1 - len),
                                                   cap - len, lane, nullptr, 0);
            if (st >= 3) { pend = st; st = 0; }                  // left the strip upwards: state pending
        }
        // (5) the token of this iteration
        TbTok N;
        N.x = x; N.y = y; N.st = st; N.len = len; N.pend = pend;
        N.job = tb_job_at(x, y);
        N.flags = (first ? 1 : 0) | (probe ? 2 : 0) | ((N.job.s < 0 || i >= max_iter) ? 4 : 0);
        if (N.flags & 4) {                                       // the walk is over: boundary runs and the length
            while (y > 0) { if (lane == 0) ops[cap - 1 - len] = 2; ++len; --y; }
            while (x > 0) { if (lane == 0) ops[cap - 1 - len] = 1; ++len; --x; }
            if (lane == 0) a.ops_len[p] = len;
        }
        if (lane == 0) {
            int* tk = tok_s[i % NWV];
            tk[0] = N.x; tk[1] = N.y; tk[2] = N.st; tk[3] = N.len; tk[4] = N.pend; tk[5] = N.flags;
            tk[6] = N.job.s; tk[7] = N.job.ck; tk[8] = N.job.gtop; tk[9] = N.job.tops;
            __hip_atomic_store(&seq_s, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (N.flags & 4) break;
        T = N;
        kt = i;
    }
}

// ---------------------------------------------------------------------------------------------
// phase 2 for LARGE batches: TWO problems per wave, each in 32 lanes, walking back HALF-strips.
//
// The one-wave kernel re-fills whole 64-lane chunks although a walk enters a chunk at some lane l and only lanes
// <= l matter: every strip costs 256 + 63 skewed steps of a full wave.  With lane 31's bottom row kept by phase 1
// (Ws2, kSubRows = 2) a strip falls into two half-strips of 128 rows that restart independently -- from the same
// state checkpoints (lanes 0 .. 31 or 32 .. 63 of them) and from the row above the half -- and a half-strip costs
// 128 + 31 steps of HALF a wave.  So a wave carries two problems, one per 32 lanes, each running the one-wave
// kernel's loop (set-up, tagged re-fill of a chunk, walk, next chunk) on its own half-strips; the two halves share
// the instruction stream and diverge only in trip counts (the compiler's EXEC masking: the half with the shorter
// re-fill or walk waits for the other).  Per problem ~0.58 of the re-filled wave-groups of the one-wave kernel.
// Everything that was wave-uniform there (position, state, chunk, pointers) is per lane here, equal within a half;
// ballots are split per half; a half's LDS arrays are its own.  Results are the one-wave kernel's, bit for bit.
constexpr int kHalfLanes = 32;
static_assert(kSubRows == 2 && kSubLanes == kHalfLanes, "nw_trace2h_kernel restarts half-strips: phase 1 must keep lane 31's rows");

template <bool CARRIED, bool SAMEGO>
__device__ __forceinline__ void refill_half(const CellRegs& kr, int (&D)[4], int (&V)[4], int (&H)[4], int& dsave,
                                            const int (&tc)[4], const int2* hvt, const uint16_t* ow, uint4* win,
                                            int2* hvb, int g0, int g_top, int m, int lam, int lb, bool lane_has_rows,
                                            int top_steps) {
    constexpr int R = 4, SPG = 4, LW = kHalfLanes;
    const int k0 = g0 * SPG;
    const bool first_lane = lam == 0;
    int oc_next[SPG];
    int2 hd_next[SPG];
    auto load_group = [&](int g) {
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            const int kk = g * SPG + q;
            oc_next[q] = ow[kk - k0 + (LW - 1) - lam];
            hd_next[q] = hvt[min(kk + 1, m + lb) - k0];          // column j = kk + 1 - lb of the row above, clamped to m
        }
    };
    auto cell = [&](int d_ul, int x_u, int y_l, int t, int o, int& d, int& x, int& y) -> unsigned {
        if constexpr (CARRIED) return cell_carried_tagged_hw<SAMEGO>(kr, d_ul, x_u, y_l, t, o, d, x, y);
        else return cell_hw(kr, d_ul, x_u, y_l, t, o, d, x, y);
    };
    // values from the lane above; the first lane of a half takes the row above the half-strip (lane 32 must not see lane 31)
    auto shift_in = [&](int& v_up, int v_src, int& d_next, int d_src) {
        const int hv = v_up, hd = d_next;
        wave_shr1_pair_sched(v_up, v_src, d_next, d_src);
        v_up = first_lane ? hv : v_up;
        d_next = first_lane ? hd : d_next;
    };
    load_group(g0);
    const int gs_lo = (63 + SPG - 1) / SPG;                   // from here on every lane of the strip has started
    auto steady_group = [&](int g, const int (&oc)[SPG], const int2 (&hd)[SPG], unsigned (&acc)[4], auto nq_c) {
        constexpr int NQ = decltype(nq_c)::value;
        int2 cap[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int v_up = hd[q].x, d_next = hd[q].y;
            shift_in(v_up, V[R - 1], d_next, D[R - 1]);
            int d_ul = dsave, v_u = v_up;
            unsigned b[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int d_old = D[rr];
                b[rr] = cell(d_ul, v_u, H[rr], tc[rr], oc[q], D[rr], V[rr], H[rr]);
                d_ul = d_old;
                v_u = V[rr];
            }
            acc[q] = pack4(b[0], b[1], b[2], b[3]);
            dsave = d_next;
            cap[q] = make_int2(V[R - 1], D[R - 1]);
        }
        if (lam == LW - 1) {                                  // the half-strip's bottom row, tagged: for pending states
#pragma unroll
            for (int q = 0; q < NQ; ++q) hvb[g * SPG - k0 + q] = cap[q];
        }
    };
    auto edge_group = [&](int g, const int (&oc)[SPG], const int2 (&hd)[SPG], unsigned (&acc)[4]) {
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            const int kk = g * SPG + q;
            const int j = kk - (lb + lam) + 1;
            const bool active = (j >= 1) && (j <= m) && lane_has_rows;
            int v_up = hd[q].x, d_next = hd[q].y;
            shift_in(v_up, V[R - 1], d_next, D[R - 1]);
            if (active) {
                int d_ul = dsave, v_u = v_up;
                unsigned b[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int d_old = D[rr];
                    b[rr] = cell(d_ul, v_u, H[rr], tc[rr], oc[q], D[rr], V[rr], H[rr]);
                    d_ul = d_old;
                    v_u = V[rr];
                }
                acc[q] = pack4(b[0], b[1], b[2], b[3]);
                dsave = d_next;
                if (lam == LW - 1) hvb[kk - k0] = make_int2(V[R - 1], D[R - 1]);
            }
        }
    };
    for (int g = g0; g <= g_top; ++g) {
        int oc[SPG];
        int2 hd[SPG];
#pragma unroll
        for (int q = 0; q < SPG; ++q) { oc[q] = oc_next[q]; hd[q] = hd_next[q]; }
        unsigned acc[4] = {0u, 0u, 0u, 0u};
        if (g < g_top) {
            load_group(g + 1);
            if (g >= gs_lo) steady_group(g, oc, hd, acc, std::integral_constant<int, SPG>{});
            else edge_group(g, oc, hd, acc);
        } else if (g >= gs_lo) {
            if (top_steps <= 2) steady_group(g, oc, hd, acc, std::integral_constant<int, 2>{});
            else steady_group(g, oc, hd, acc, std::integral_constant<int, SPG>{});
        } else {
            edge_group(g, oc, hd, acc);
        }
        win[(g - g0) * LW + lam] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// walk_window_vec (nw_hw.h) for one half of a wave: 32 lanes look ahead along the run, the ballots are split per half,
// position and state are per-lane copies.  win: [(group - gw_lo) * 32 + (strip lane - lb)]; x_lo: the row above the
// half-strip.  A step that leaves a half-strip below the table's first upwards ends the walk with st = 3 + state.
__device__ __forceinline__ int walk_half(const uint4* win, int gw_lo, int klow, int x_lo, int lb, int& x, int& y, int& st,
                                         uint8_t* opsbuf, int max_ops, int lam, int half) {
    constexpr int R = 4, LW = kHalfLanes;
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
    int cnt = 0;
    while (true) {
        const int up = (st != 2), left = (st != 1);
        const int xi = x - lam * up, yi = y - lam * left;
        const int li = ((xi - 1 - x_lo) >> 2) + lb;                       // strip lane of the cell this lane looks at
        const int ki = (yi - 1) + li;
        const bool valid = (xi > x_lo) & (yi > 0) & (ki >= klow);
        unsigned b = 0;
        if (valid) b = wb[((ki >> 2) - gw_lo) * (LW * 16) + (li - lb) * 16 + (ki & 3) * R + ((xi - 1) & (R - 1))];
        int nxt = 2 - (int)((b >> (2 * st)) & 3u);
        if (up && x_lo > 0 && xi == x_lo + 1) nxt = 3 + st;
        const unsigned long long vm64 = __ballot(valid);
        const unsigned long long cm64 = __ballot(valid && nxt == st);
        const unsigned vmask = half ? (unsigned)(vm64 >> 32) : (unsigned)vm64;
        const unsigned cmask = half ? (unsigned)(cm64 >> 32) : (unsigned)cm64;
        if ((vmask & 1u) == 0u) break;                                     // the current cell is out
        const int run = (~cmask == 0u) ? LW : (int)__builtin_ctz(~cmask);  // lanes 0 .. run - 1 stay in st
        int steps = run, st_new = st;
        // (an LDS shuffle: every lane of the half asks the same lane.  Two scalar v_readlane per half instead -- each
        // half's run length is uniform in the half -- measured slower, 2.46 against 2.42 ms: the scalar round trip
        // stalls the wave longer than the ds_bpermute it would save)
        const int nxt_at = __shfl(nxt, half * LW + min(run, LW - 1), 64);
        if (run < LW && ((vmask >> run) & 1u)) {                           // the step that leaves state st
            steps = run + 1;
            st_new = nxt_at;
        }
        steps = min(steps, max_ops - cnt);
        if (steps < run + 1) st_new = st;                                  // truncated inside the run
        if (lam < steps) opsbuf[-(cnt + lam)] = (uint8_t)st;
        cnt += steps;
        x -= steps * up;
        y -= steps * left;
        st = st_new;
        if (cnt >= max_ops) break;
    }
    return cnt;
}

__global__ __launch_bounds__(64) void nw_trace2h_kernel(NwArgs a) {
    constexpr int R = 4, LW = kHalfLanes, SRH = LW * R;        // 128 rows per half-strip
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    __shared__ uint4 win_s[2][kChunkGroups * LW];
    __shared__ int2 hvt_s[2][kChunkSteps + 8];
    __shared__ int2 hvb_s[2][kChunkSteps];
    __shared__ uint16_t ow_s[2][kChunkSteps + LW + 8];

    const int lane = threadIdx.x, half = lane >> 5, lam = lane & (LW - 1);
    const int pr = blockIdx.x * 2 + half;
    const bool alive = pr < a.nprob;
    const int p = alive ? pr : a.nprob - 1;                     // (an odd batch: the last wave's second half idles)
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = alive ? (int)(a.t_off[p + 1] - t0) : 0;
    const int m = alive ? (int)(a.o_off[p + 1] - o0) : 0;
    uint8_t* const ops = a.ops_out + a.ops_off[p];
    const int cap = n + m;
    uint4* const win = win_s[half];
    int2* const hvt = hvt_s[half];
    int2* const hvb = hvb_s[half];
    uint16_t* const ow = ow_s[half];
    // a half's LDS writes followed by its own lanes' reads: the LDS executes a wave's operations in order; this only
    // keeps the compiler from moving them across
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int x = n, y = m, len = 0, st = 0, pend = 0;
    bool first = true, probe = false;
    if (n > 1 && (n - 1) % SRH == 0) {                          // the start state PM(n, m) is a tag of the half-strip above
        if (m == 1) { st = 0; first = false; }
        else { x = n - 1; y = m - 1; pend = 3; probe = true; }
    }
    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    CellRegs kr;
    kr.cmis = c.cmismatch; kr.cmat = c.cmatch; kr.gox6 = c.gox6; kr.goy6 = c.goy6;
    kr.clean = ~kTagMask;
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int xadj = carried ? c.gox : 0, yadj = carried ? c.goy : 0;
    const int xadj6 = xadj * 64, yadj6 = yadj * 64;
    const Ws2 ws(max(n, 1), max(m, 1));
    uint8_t* const ws_p = a.ws + a.ws_off[p];
#if TA_P2_PROFILE
    // (wave-level clocks: the halves move in lockstep, so a phase's time is the slower half's; pc_groups / pc_iters
    // count THIS half's own re-filled groups and walk iterations)
    long long pc_setup = 0, pc_fill = 0, pc_walk = 0, pc_chunks = 0, pc_t = __builtin_readcyclecounter(), pc_groups = 0;
    const long long pc_start = pc_t;
    long long pc_iters = 0;
#endif

    while (x > 0 && y > 0) {
        const int hs = (x - 1) / SRH;                           // half-strip the walk is in: strip hs / 2, lanes lb ..
        const int s = hs >> 1, lb = (hs & 1) * LW;
        int l = ((x - 1) % L::SR) / R;                          // strip lane
        int r = (x - 1) % R;
        int k = (y - 1) + l;
        const int i_h = hs * SRH;                               // 1-based index of the row above the half-strip
        const int row0 = s * L::SR + (lb + lam) * R;
        const bool lane_has_rows = row0 < n;
        int tc[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = row0 + rr + 1;
            tc[rr] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
        }
        const int2* const hrow = reinterpret_cast<const int2*>(ws_p + ws.row(hs)) + 1;    // the row above: entry j

        int ck = (k >> 2) / kChunk;
        int g_top = k >> 2;
        if (ck > 0 && k < ck * kChunk * SPG + 2) ck -= 1;       // a chunk's first two steps belong to the chunk before
        bool in_strip = true;
        constexpr int kOwIt = (kChunkSteps + LW + LW - 1) / LW, kRowIt = (kChunkSteps + 1 + LW - 1) / LW;
        int in_ow[kOwIt], in_st[kStateInts];
        int2 in_row[kRowIt];
        int in_ck = -1, in_gtop = -1;                           // what the registers hold (this half-strip)
        auto fetch_inputs = [&](int ck_, int gtop_) {
            const int g0_ = ck_ * kChunk, k0_ = g0_ * SPG;
            const int nsteps_ = (gtop_ - g0_ + 1) * SPG;
#pragma unroll
            for (int it = 0; it < kOwIt; ++it) {
                const int src = k0_ - lb - (LW - 1) + it * LW + lam;
                in_ow[it] = (src >= 0 && src < m) ? a.o_codes[o0 + src] : 0xFFFF;
            }
            const int jlo_ = k0_ - lb, jhi_ = min(m, jlo_ + nsteps_);
#pragma unroll
            for (int it = 0; it < kRowIt; ++it) {
                const int jj = min(jlo_ + it * LW + lam, jhi_);
                if (hs == 0) in_row[it] = make_int2(bnd_V_row0(c, max(jj, 0)) + xadj6, bnd_D_row0(c, max(jj, 0)));
                else {
                    const int2 e = hrow[max(jj, 1)];
                    in_row[it] = (jj <= 0) ? make_int2(0, bnd_D_col0(c, i_h)) : make_int2(enc_of(e.x), enc_of(e.y));
                }
            }
            if (g0_ > 0) {
                const int* stp = reinterpret_cast<const int*>(ws_p + ws.state(s, g0_ / kCkGroups)) + lb + lam;
#pragma unroll
                for (int q = 0; q < kStateInts; ++q) in_st[q] = stp[q * 64];
            }
            in_ck = ck_; in_gtop = gtop_;
        };
        while (in_strip) {
            const int g0 = ck * kChunk;
            const int k0 = g0 * SPG;
            const int kvalid = ck > 0 ? k0 + 2 : 0;
            const int nsteps_w = (g_top - g0 + 1) * SPG;
            // Inputs of the re-fill, fetched into registers (fetch_inputs): (a) OCR codes ow[i] = o[(k0 - lb - 31) + i],
            // (b) the row above, hvt[jj - (k0 - lb)] for columns jj, (c) the lane state at group g0.  The chunk after
            // this one is nearly always the one before it in the same half-strip, whole: its inputs are requested as
            // soon as this chunk's are in LDS and arrive under this chunk's re-fill and walk.
            if (in_ck != ck || in_gtop != g_top) fetch_inputs(ck, g_top);
#pragma unroll
            for (int it = 0; it < kOwIt; ++it) {
                const int i = it * LW + lam;
                if (i < nsteps_w + LW) ow[i] = (uint16_t)in_ow[it];
            }
            const int jlo = k0 - lb, jhi = min(m, jlo + nsteps_w);
#pragma unroll
            for (int it = 0; it < kRowIt; ++it) {
                const int jj = jlo + it * LW + lam;
                if (jj <= jhi) hvt[jj - jlo] = in_row[it];
            }
            int D[R], V[R], H[R], dsave;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int i = row0 + rr + 1;
                V[rr] = 0;
                D[rr] = bnd_D_col0(c, i);
                H[rr] = bnd_H_col0(c, i) + yadj6;
            }
            dsave = bnd_D_col0(c, row0);
            if (g0 > 0 && lb + lam < k0) {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) { D[rr] = enc_of(in_st[rr]); H[rr] = enc_of(in_st[R + rr]); }
                V[R - 1] = enc_of(in_st[2 * R]);
                dsave = enc_of(in_st[2 * R + 1]);
            }
            if (ck > 0) fetch_inputs(ck - 1, g0);              // the likely next chunk (the registers are free again)
            wave_sync();
            PC_LAP(pc_setup)
            // (d) tagged re-fill of groups g0 .. g_top
            {
                const int top_steps = ((k >> 2) == g_top) ? (k & 3) + 1 : SPG;
                if (carried && c.gox == c.goy) refill_half<true, true>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, top_steps);
                else if (carried) refill_half<true, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, top_steps);
                else refill_half<false, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, top_steps);
            }
            wave_sync();
            PC_LAP(pc_fill)
#if TA_P2_PROFILE
            pc_chunks += 1; pc_groups += g_top - g0 + 1;
#endif
            // (e) walk the chunk
            const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
            if (pend) {                                        // (x, y): last row of this half-strip, step k of this chunk
                const int2 e = hvb[k - k0];
                st = 2 - (((pend == 3) ? e.y : e.x) & 3);
                pend = 0;
                if (probe) {                                   // that was the start state: back to (n, m)
                    probe = false; first = false;
                    x = n; y = m;
                    break;
                }
            }
            if (first && k >= kvalid) {                        // start state, textSeqCompare.py:102
                st = ptr_pm(wb[(((k >> 2) - g0) * LW + (l - lb)) * 16 + (k & 3) * R + r]);
                first = false;
            }
            len += walk_half(win, g0, kvalid, i_h, lb, x, y, st, ops + (cap - 1 - len), cap - len, lam, half);
            if (st >= 3) { pend = st; st = 0; }                // left the half-strip upwards: state pending
            PC_LAP(pc_walk)
            l = (x > i_h) ? ((x - 1) % L::SR) / R : -1;
            r = (x - 1) & (R - 1);
            k = (y - 1) + l;
            if ((x <= 0) | (y <= 0) | (l < lb)) {
                in_strip = false;                              // the walk left the half-strip or finished
            } else if (k >= kvalid) {
                g_top = k >> 2;                                // (not reached: whole half-strip chunks are kept)
            } else {
                g_top = g0;                                    // the chunk before, through its halo
                ck -= 1;
            }
        }
    }
    while (y > 0) { if (lam == 0) ops[cap - 1 - len] = 2; ++len; --y; }
    while (x > 0) { if (lam == 0) ops[cap - 1 - len] = 1; ++len; --x; }
    if (alive && lam == 0) a.ops_len[p] = len;
#if TA_P2_PROFILE
    if (alive && lam == 0) {                           // row 0 of the workspace is phase 1's: free by now
        long long* out = reinterpret_cast<long long*>(ws_p + ws.row(0));
        out[0] = pc_setup; out[1] = pc_fill; out[2] = pc_walk; out[3] = pc_chunks; out[4] = pc_groups; out[5] = len;
        out[6] = pc_iters; out[7] = __builtin_readcyclecounter() - pc_start;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// phase 2 for MEDIUM batches: both of the above at once -- two problems per wave on half-strips (nw_trace2h_kernel:
// ~0.58 of the re-filled cells) AND several waves per pair of problems that re-fill the chunks they expect ahead of
// the walk (nw_trace2w_kernel: the chain of set-up -> re-fill -> walk off the critical path).  A workgroup of NWV
// waves owns two problems, one per 32 lanes of every wave; iteration i -- one chunk job of each problem -- belongs to
// wave i mod NWV, which re-fills the jobs it expects for BOTH halves into its own LDS windows, waits for the tokens of
// iteration i - 1 (one per half), re-fills again the halves whose expectation was wrong, walks both and passes the
// tokens on.  A half whose problem is finished idles (its token says so); the workgroup leaves when both are.
struct HJob { int hs, ck, gtop, tops; };                       // half-strip, chunk, last group, steps of it; hs < 0: none
__device__ __forceinline__ HJob hjob_at(int x, int y) {
    if (x <= 0 || y <= 0) return HJob{-1, 0, 0, 0};
    const int l = ((x - 1) % 256) / 4, k = (y - 1) + l;
    int ck = (k >> 2) / kChunk;
    if (ck > 0 && k < ck * kChunk * 4 + 2) ck -= 1;              // a chunk's first two steps belong to the chunk before
    return HJob{(x - 1) / (kHalfLanes * 4), ck, k >> 2, (k & 3) + 1};
}
__device__ __forceinline__ HJob hjob_prev(const HJob& j) {
    if (j.hs < 0 || j.ck < 1) return HJob{-1, 0, 0, 0};
    return HJob{j.hs, j.ck - 1, j.ck * kChunk, 2};
}
__device__ __forceinline__ HJob hjob_predict(int x, int y, HJob j, int d) {        // tb_predict on half-strips
    for (; d > 0; --d) {
        if (j.hs < 0) break;
        const int l = ((x - 1) % 256) / 4, k = (y - 1) + l;
        const int to_top = x - j.hs * (kHalfLanes * 4);
        const int to_halo = j.ck > 0 ? ((k - (j.ck * kChunk * 4 + 1)) * 4 + 4) / 5 : (1 << 28);
        if (to_halo < to_top) {
            x -= to_halo; y -= to_halo;
            j = (y > 0) ? hjob_prev(j) : HJob{-1, 0, 0, 0};
        } else {
            x -= to_top; y -= to_top;
            j = hjob_at(x, y);
            if (j.hs >= 0) { j.gtop = min(j.gtop + 2, j.ck * kChunk + kChunk); j.tops = 4; }
        }
    }
    return j;
}
__device__ __forceinline__ bool hjob_serves(const HJob& spec, const HJob& j) {
    return spec.hs >= 0 && spec.hs == j.hs && spec.ck == j.ck &&
           (spec.gtop > j.gtop || (spec.gtop == j.gtop && spec.tops >= j.tops));
}

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void nw_trace2hw_kernel(NwArgs a) {
    constexpr int R = 4, LW = kHalfLanes, SRH = LW * R;
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    constexpr int kTok = 10;
    __shared__ uint4 win_s[NWV][2][kChunkGroups * LW];
    __shared__ int2 hvt_s[NWV][2][kChunkSteps + 8];
    __shared__ int2 hvb_s[NWV][2][kChunkSteps];
    __shared__ uint16_t ow_s[NWV][2][kChunkSteps + LW + 8];
    __shared__ int tok_s[NWV][2][kTok];
    __shared__ int seq_s;

    const int lane = threadIdx.x & 63, half = lane >> 5, lam = lane & (LW - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pr = blockIdx.x * 2 + half;
    const bool alive = pr < a.nprob;
    const int p = alive ? pr : a.nprob - 1;
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = alive ? (int)(a.t_off[p + 1] - t0) : 0;
    const int m = alive ? (int)(a.o_off[p + 1] - o0) : 0;
    uint8_t* const ops = a.ops_out + a.ops_off[p];
    const int cap = n + m;
    uint4* const win = win_s[wave][half];
    int2* const hvt = hvt_s[wave][half];
    int2* const hvb = hvb_s[wave][half];
    uint16_t* const ow = ow_s[wave][half];
    if (threadIdx.x == 0) seq_s = 0;
    __syncthreads();                                            // the only workgroup barrier
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    CellRegs kr;
    kr.cmis = c.cmismatch; kr.cmat = c.cmatch; kr.gox6 = c.gox6; kr.goy6 = c.goy6;
    kr.clean = ~kTagMask;
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int xadj = carried ? c.gox : 0, yadj = carried ? c.goy : 0;
    const int xadj6 = xadj * 64, yadj6 = yadj * 64;
    const Ws2 ws(max(n, 1), max(m, 1));
    uint8_t* const ws_p = a.ws + a.ws_off[p];

    // this half's token before iteration 0 (the same on every wave): flags 1 first, 2 probe, 4 done
    int tx = n, ty = m, tst = 0, tlen = 0, tpend = 0, tflags = 1;
    if (n > 1 && (n - 1) % SRH == 0) {
        if (m == 1) tflags = 0;
        else { tx = n - 1; ty = m - 1; tpend = 3; tflags = 1 | 2; }
    }
    HJob tj = hjob_at(tx, ty);
    if (tj.hs < 0) {                                            // an empty string (or no problem in this half): boundary run only
        tflags = 4;
        if (wave == 0) {
            int x = n, y = m, len = 0;
            while (y > 0) { if (lam == 0) ops[cap - 1 - len] = 2; ++len; --y; }
            while (x > 0) { if (lam == 0) ops[cap - 1 - len] = 1; ++len; --x; }
            if (alive && lam == 0) a.ops_len[p] = len;
        }
    }
    if (__all((tflags & 4) != 0)) return;
    int kt = -1;

    // tagged re-fill of this half's job J (set-up + refill_half) where `on`; the other half waits
    auto refill = [&](const HJob& J, bool on) {
        if (on) {
            const int hs = J.hs, s = hs >> 1, lb = (hs & 1) * LW;
            const int g0 = J.ck * kChunk, k0 = g0 * SPG, g_top = J.gtop;
            const int nsteps_w = (g_top - g0 + 1) * SPG;
            const int i_h = hs * SRH;
            const int row0 = s * L::SR + (lb + lam) * R;
            const bool lane_has_rows = row0 < n;
            int tc[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int i = row0 + rr + 1;
                tc[rr] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
            }
            const int2* const hrow = reinterpret_cast<const int2*>(ws_p + ws.row(hs)) + 1;
            constexpr int kOwIt = (kChunkSteps + LW + LW - 1) / LW, kRowIt = (kChunkSteps + 1 + LW - 1) / LW;
#pragma unroll
            for (int it = 0; it < kOwIt; ++it) {
                const int i = it * LW + lam, src = k0 - lb - (LW - 1) + i;
                if (i < nsteps_w + LW) ow[i] = (uint16_t)((src >= 0 && src < m) ? a.o_codes[o0 + src] : 0xFFFF);
            }
            const int jlo = k0 - lb, jhi = min(m, jlo + nsteps_w);
#pragma unroll
            for (int it = 0; it < kRowIt; ++it) {
                const int jj = jlo + it * LW + lam;
                if (jj <= jhi) {
                    int2 v;
                    if (hs == 0) v = make_int2(bnd_V_row0(c, max(jj, 0)) + xadj6, bnd_D_row0(c, max(jj, 0)));
                    else {
                        const int2 e = hrow[max(jj, 1)];
                        v = (jj <= 0) ? make_int2(0, bnd_D_col0(c, i_h)) : make_int2(enc_of(e.x), enc_of(e.y));
                    }
                    hvt[jj - jlo] = v;
                }
            }
            int D[R], V[R], H[R], dsave;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int i = row0 + rr + 1;
                V[rr] = 0;
                D[rr] = bnd_D_col0(c, i);
                H[rr] = bnd_H_col0(c, i) + yadj6;
            }
            dsave = bnd_D_col0(c, row0);
            if (g0 > 0 && lb + lam < k0) {
                const int* stp = reinterpret_cast<const int*>(ws_p + ws.state(s, g0 / kCkGroups)) + lb + lam;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) { D[rr] = enc_of(stp[rr * 64]); H[rr] = enc_of(stp[(R + rr) * 64]); }
                V[R - 1] = enc_of(stp[2 * R * 64]);
                dsave = enc_of(stp[(2 * R + 1) * 64]);
            }
            wave_sync();
            if (carried && c.gox == c.goy) refill_half<true, true>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, J.tops);
            else if (carried) refill_half<true, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, J.tops);
            else refill_half<false, false>(kr, D, V, H, dsave, tc, hvt, ow, win, hvb, g0, g_top, m, lam, lb, lane_has_rows, J.tops);
            wave_sync();
        }
    };

    const int max_iter = 8 * (2 * L::nstrips(max(n, 1)) * (ws.ngroups / kChunk + 2)) + 64;
    const int max_iter_w = __builtin_amdgcn_readfirstlane(max(__shfl(max_iter, 0, 64), __shfl(max_iter, LW, 64)));
    for (int i = wave; ; i += NWV) {
        // (1) speculation + (2) the tokens of iteration i - 1
        HJob spec{-1, 0, 0, 0};
        if (i > kt + 1) {
            int spins = 0, based_on = -2;
            while (true) {
                const int have = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&seq_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (have >= i) break;
                const int from = max(kt, have - 1);
                if (from != based_on) {
                    based_on = from;
                    HJob want = tj;
                    int px = tx, py = ty;
                    bool over = tflags & 4;
                    if (from != kt) {
                        const int* tq = tok_s[from % NWV][half];
                        want = HJob{tq[6], tq[7], tq[8], tq[9]};
                        px = tq[0]; py = tq[1];
                        over = tq[5] & 4;
                    }
                    if (over) want = HJob{-1, 0, 0, 0};
                    else want = hjob_predict(px, py, want, i - from - 1);
                    const bool change = want.hs >= 0 &&
                        !(spec.hs >= 0 && want.hs == spec.hs && want.ck == spec.ck && want.gtop == spec.gtop);
                    if (__any(change)) {
                        if (change) spec = want;
                        refill(spec, change);
                        continue;
                    }
                }
                if (++spins >= (1 << 24)) break;                 // (bounded: a mistake could never leave waves spinning)
                __builtin_amdgcn_s_sleep(4);
            }
            if (spins >= (1 << 24)) break;
            const int* tk = tok_s[(i - 1) % NWV][half];
            tx = tk[0]; ty = tk[1]; tst = tk[2]; tlen = tk[3]; tpend = tk[4]; tflags = tk[5];
            tj = HJob{tk[6], tk[7], tk[8], tk[9]};
            kt = i - 1;
        }
        if (__all((tflags & 4) != 0)) {                          // both walks are over: pass the word on and leave
            if (lam == 0) tok_s[i % NWV][half][5] = 4;
            if (lane == 0) __hip_atomic_store(&seq_s, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        const bool live = (tflags & 4) == 0;                     // this half still walks
        // (3) the chunk itself where the speculation was wrong
        const HJob J = tj;
        const bool need = live && !hjob_serves(spec, J);
        if (__any(need)) refill(J, need);
        // (4) walk (nw_trace2h_kernel, step (e))
        int x = tx, y = ty, st = tst, len = tlen, pend = tpend;
        bool first = tflags & 1, probe = tflags & 2;
        if (live) {
            const int hs = J.hs, lb = (hs & 1) * LW, i_h = hs * SRH;
            const int g0 = J.ck * kChunk, k0 = g0 * SPG;
            const int kvalid = J.ck > 0 ? k0 + 2 : 0;
            const int l = ((x - 1) % L::SR) / R, r = (x - 1) % R, k = (y - 1) + l;
            bool walked = true;
            if (pend) {
                const int2 e = hvb[k - k0];
                st = 2 - (((pend == 3) ? e.y : e.x) & 3);
                pend = 0;
                if (probe) { probe = false; first = false; x = n; y = m; walked = false; }
            }
            if (walked) {
                if (first && k >= kvalid) {
                    st = ptr_pm(reinterpret_cast<const uint8_t*>(win)[(((k >> 2) - g0) * LW + (l - lb)) * 16 + (k & 3) * R + r]);
                    first = false;
                }
                len += walk_half(win, g0, kvalid, i_h, lb, x, y, st, ops + (cap - 1 - len), cap - len, lam, half);
                if (st >= 3) { pend = st; st = 0; }
            }
        }
        // (5) the tokens of this iteration
        HJob nj = live ? hjob_at(x, y) : HJob{-1, 0, 0, 0};
        int nflags = (first ? 1 : 0) | (probe ? 2 : 0) | ((!live || nj.hs < 0 || i >= max_iter_w) ? 4 : 0);
        if (live && (nflags & 4)) {                              // this half's walk ends here: boundary runs and the length
            while (y > 0) { if (lam == 0) ops[cap - 1 - len] = 2; ++len; --y; }
            while (x > 0) { if (lam == 0) ops[cap - 1 - len] = 1; ++len; --x; }
            if (alive && lam == 0) a.ops_len[p] = len;
        }
        if (lam == 0) {
            int* tk = tok_s[i % NWV][half];
            tk[0] = x; tk[1] = y; tk[2] = st; tk[3] = len; tk[4] = pend; tk[5] = nflags;
            tk[6] = nj.hs; tk[7] = nj.ck; tk[8] = nj.gtop; tk[9] = nj.tops;
        }
        if (lane == 0) __hip_atomic_store(&seq_s, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__all((nflags & 4) != 0)) break;
        tx = x; ty = y; tst = st; tlen = len; tpend = pend; tflags = nflags; tj = nj;
        kt = i;
    }
}

}  // namespace ta

using namespace ta;

// widest problem of the two-phase aligner: its LDS holds the OCR codes (2 B each) only
extern "C" int32_t ta_nw2_max_m(void) { return 65000; }

extern "C" int64_t ta_nw2_workspace_bytes(int32_t n, int32_t m) {
    if (n <= 0 || m <= 0) return 16;
    return (Ws2(n, m).total + 15) & ~(int64_t)15;
}

template <int W, int MODE, bool SAMEGO, typename OC>
static hipError_t launch_score_k(const NwArgs& a, size_t lds, hipStream_t st) {
    hipError_t e = allow_full_lds(&nw_score_kernel<W, MODE, SAMEGO, OC>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((nw_score_kernel<W, MODE, SAMEGO, OC>), dim3(a.nprob), dim3(W * 64), lds, st, a);
    return hipGetLastError();
}

template <int W>
static hipError_t launch_score_w(const NwArgs& a, size_t lds, bool profile, bool samego, bool codes8, hipStream_t st) {
    if (profile) return samego ? launch_score_k<W, 2, true, uint16_t>(a, lds, st)
                               : launch_score_k<W, 2, false, uint16_t>(a, lds, st);
    return codes8 ? launch_score_k<W, 1, false, uint8_t>(a, lds, st)
                  : launch_score_k<W, 1, false, uint16_t>(a, lds, st);
}

// Phase-1 launch shape.  Waves per workgroup W <= strips of the tallest problem; with the score
// profile the LDS per workgroup is W x apad x 256 B + the codes.  W is chosen by a small model fitted
// to measurements (tools/p1_time.py on 2048 x 1024^2, 1024 / 2048 x 2048^2, 4096 x 4096^2,
// 512 x 8192^2: it picks the fastest W in each): score = min(resident waves per CU, 12)
// x (share of a workgroup's time that is not start-up ramp) x (share of the chip the batch fills).
// What workgroups that share a CU can hold together.  One workgroup may be given 160 KiB, but measured on
// MI355X (nw_score_kernel with its LDS request padded, tools/p1_time.py with TA_NW2_LDS_PAD): three
// workgroups of 41.3 KiB and two of 53.3 KiB run side by side, two of 63.3 KiB do not -- the limit for
// co-residency lies between 124 and 126 KiB.  (The single-wave workgroups of nw_trace2_kernel, 19.4 KiB
// static each, do run eight to a CU; the figure below is for this kernel's launch shapes.)
constexpr size_t kLdsShared = 124 * 1024;
struct P1Plan { int mode, w, apad; size_t lds; bool samego, codes8; };

static P1Plan plan_phase1(int max_n, int max_m, uint32_t flags, int nprob = 1 << 20) {
    P1Plan pl{};
    const int nstrips = PtrLayout<4>::nstrips(max_n);
    const int wmax = nstrips >= 8 ? 8 : nstrips >= 4 ? 4 : nstrips >= 2 ? 2 : 1;
    pl.codes8 = (flags & TA_NW_CODES8) != 0;
    pl.samego = (flags & TA_NW_OPENS_SAME) != 0;
    const int alphabet = (int)((flags >> TA_NW_ALPHABET_SHIFT) & 0xFFu);
    bool profile = alphabet > 0 && alphabet < 255 && !(flags & TA_NW_NO_PROFILE);
    pl.w = wmax;
    const int ngroups = PtrLayout<4>::ngroups(max_m);
    // best W for a workgroup that needs lds(W) bytes; returns the resident waves of the choice
    auto choose = [&](auto lds_of, int w_least, int& w_out) -> int {
        int best_w = 0, best_res = 0;
        double best_score = 0.0;
        for (int cand : {4, 8, 2, 1}) {            // order of preference among equals (measured: 4 >= 8 > 2)
            if (cand > wmax || cand < w_least) continue;
            const size_t need = lds_of(cand);
            if (need > 160 * 1024) continue;
            // resident waves per CU: whole workgroups, within the LDS and within 5 waves per SIMD (VGPRs);
            // beyond three per SIMD the kernel gains nothing (it is bound by VALU issue from there on)
            const int res = (int)std::min<size_t>(kLdsShared / need, 20 / cand) * cand;
            // the waves of a workgroup start 25 groups apart: with few passes over the strips that
            // ramp is a visible share of a workgroup's time (W = 8 on 8 strips of 2048 columns: 25 %)
            const int passes = (nstrips + cand - 1) / cand;
            const double busy = (double)passes * ngroups / ((double)passes * ngroups + 25.0 * (cand - 1));
            // and the batch has to fill the chip: nprob x W waves for 256 CUs x 12
            const double fill = std::min(1.0, (double)nprob * cand / (256.0 * 12.0));
            const double score = std::min(res, 12) * busy * fill;
            if (score > best_score + 1e-9) { best_score = score; best_res = res; best_w = cand; }
        }
        w_out = best_w;
        return best_res;
    };
    if (profile) {
        pl.apad = alphabet + 1;
        int w = 0;
        const int res = choose([&](int cand) { return P1Lds(max_m, 2, cand * pl.apad * 256).total; }, 1, w);
        // a profile that leaves fewer than 8 waves on a CU is not worth its LDS
        if (w == 0 || res < 8) profile = false;
        else pl.w = w;
    }
    if (!profile) {
        int w = 0;
        // compare-select cell: measured 4 >= 8 > 2 at 4096 x 4096^2 (14.5 / 14.9 / 16.9 ms); narrower
        // workgroups were not fitted for this mode
        choose([&](int) { return P1Lds(max_m, pl.codes8 ? 1 : 2).total; }, std::min(4, wmax), w);
        if (w > 0) pl.w = w;
    }
    if (const int v = (int)((flags >> TA_NW_WAVES_SHIFT) & 0xFu)) {      // caller fixes the width (tests / tuning)
        if ((v == 1 || v == 2 || v == 4 || v == 8) && v <= wmax &&
            (!profile || P1Lds(max_m, 2, v * pl.apad * 256).total <= 160 * 1024))
            pl.w = v;
    }
    pl.mode = profile ? 2 : 1;
    if (!profile) pl.apad = 0;
    pl.lds = profile ? P1Lds(max_m, 2, pl.w * pl.apad * 256).total : P1Lds(max_m, pl.codes8 ? 1 : 2).total;
    return pl;
}

extern "C" int ta_nw2_phase1_plan_batch(int32_t max_n, int32_t max_m, int32_t nprob, uint32_t flags, int32_t* out) {
    if (!out || max_n < 0 || max_m < 0 || nprob < 0) return ta_fail(TA_EINVAL, "bad argument");
    const P1Plan pl = plan_phase1(max_n, max_m, flags, nprob);
    out[0] = pl.mode; out[1] = pl.w; out[2] = (int32_t)pl.lds; out[3] = pl.samego ? 1 : 0;
    return TA_OK;
}

extern "C" int ta_nw2_phase1_plan(int32_t max_n, int32_t max_m, uint32_t flags, int32_t* out) {
    return ta_nw2_phase1_plan_batch(max_n, max_m, 1 << 20, flags, out);
}

static hipError_t launch_score(NwArgs a, int max_n, int max_m, uint32_t flags, hipStream_t st) {
    const P1Plan pl = plan_phase1(max_n, max_m, flags, a.nprob);
    if (pl.lds > 160 * 1024) return hipErrorInvalidValue;
    a.apad = pl.apad;
    switch (pl.w) {
    case 8: return launch_score_w<8>(a, pl.lds, pl.mode == 2, pl.samego, pl.codes8, st);
    case 4: return launch_score_w<4>(a, pl.lds, pl.mode == 2, pl.samego, pl.codes8, st);
    case 2: return launch_score_w<2>(a, pl.lds, pl.mode == 2, pl.samego, pl.codes8, st);
    default: return launch_score_w<1>(a, pl.lds, pl.mode == 2, pl.samego, pl.codes8, st);
    }
}

// Phase-2 launch shape: 1, 2 or 4 = waves per problem (nw_trace2_kernel / nw_trace2w_kernel), 3 = two problems per wave
// on half-strips (nw_trace2h_kernel), 5 / 6 = two / four waves per PAIR of problems on half-strips
// (nw_trace2hw_kernel).  By batch size, from measurements at 2048^2 (tools/tb_waves_time.py,
// profiles/r04_traceback_waves_per_problem.txt; ms with 1 / 2 / 4 waves / pairs / pairs x 2 / pairs x 4):
//    64 x 0.63 / 0.36 / 0.22 / 0.86 /  --  /  --        768 x 0.66 / 0.50 / 0.48 / 0.87 / 0.66 / 0.46
//  1024 x 0.67 / 0.52 / 0.56 / 0.87 / 0.67 / 0.48      1280 x 0.91 / 0.78 / 0.72 / 0.88 / 0.68 / 0.73
//  1536 x 0.91 / 0.97 / 0.84 / 0.89 / 0.68 / 0.73      2048 x 0.93 / 1.02 / 1.11 / 0.91 / 0.72 / 0.90
//  3072 x 1.21 / 1.51 / 1.65 / 1.23 / 1.34 / 1.34      4096 x 1.49 / 2.06 / 2.16 / 1.24 / 1.41 / 1.75
// and 4096 x 4096^2 2.88 / 3.93 / 4.31 / 2.45 / 2.86 / 3.55 (four-wave workgroups hold 75 - 81 KB of LDS: two per CU).
// The pair kernels only when the batch shares one scoring system (their halves run in lockstep, and problems scored
// differently have paths of very different length).
extern "C" int32_t ta_nw2_traceback_plan(int32_t nprob, int32_t params_stride, uint32_t flags) {
    const int tbw = (int)((flags >> TA_NW_TBWAVES_SHIFT) & 0x7u);
    if (tbw >= 1 && tbw <= 6) return tbw;
    if (nprob <= 832) return 4;
    if (params_stride != 0) return nprob <= 1088 ? 2 : nprob <= 1600 ? 4 : 1;
    if (nprob <= 1088) return 6;
    if (nprob <= 2560) return 5;
    return 3;
}

// TA_NW_CHECK_IDS: every token id of the batch against the bound the caller's flags assert (one workgroup per problem)
__global__ __launch_bounds__(256) void nw_check_ids_kernel(NwArgs a, int bound, int* bad) {
    const int p = blockIdx.x;
    int mine = 0;
    for (int64_t i = a.t_off[p] + threadIdx.x; i < a.t_off[p + 1]; i += 256) mine |= (unsigned)a.t_codes[i] >= (unsigned)bound;
    for (int64_t i = a.o_off[p] + threadIdx.x; i < a.o_off[p + 1]; i += 256) mine |= (unsigned)a.o_codes[i] >= (unsigned)bound;
    if (__ballot(mine) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

static int check_ids(const NwArgs& a, uint32_t flags, hipStream_t st) {
    int bound = 65535;                                        // the ABI's own limit (ids < 65536; 65535 is the pad code)
    if (flags & TA_NW_CODES8) bound = 255;
    const int alpha = (int)((flags >> TA_NW_ALPHABET_SHIFT) & 0xFFu);
    if (alpha > 0 && !(flags & TA_NW_NO_PROFILE)) bound = alpha < bound ? alpha : bound;
    // stream-ordered allocation: hipMalloc / hipFree would synchronise the whole DEVICE (the page pipeline's other streams
    // included), this call waits for `st` alone
    int* bad = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&bad), sizeof(int), st);
    if (e != hipSuccess) return ta_fail_hip(e, "TA_NW_CHECK_IDS: hipMallocAsync");
    int host_bad = 0;
    e = hipMemsetAsync(bad, 0, sizeof(int), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(nw_check_ids_kernel, dim3(a.nprob), dim3(256), 0, st, a, bound, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
    (void)hipFreeAsync(bad, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return ta_fail_hip(e, "TA_NW_CHECK_IDS");
    if (host_bad) return ta_fail(TA_EINVAL, "a token id is outside the range the flags assert (TA_NW_CODES8 / TA_NW_ALPHABET)");
    return TA_OK;
}

extern "C" int ta_nw2_batch(const int32_t* t_codes, const int64_t* t_off,
                            const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                            const int32_t* params, int32_t params_stride,
                            uint8_t* ws, const int64_t* ws_off,
                            uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len,
                            int32_t max_n, int32_t max_m, int64_t score_bound,
                            uint32_t flags, void* stream) {
    if (nprob < 0 || max_n < 0 || max_m < 0) return ta_fail(TA_EINVAL, "negative size");
    if (nprob == 0) return TA_OK;
    if (!t_off || !o_off || !params || !ws_off || !ops_off || !ops_len)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if (params_stride != 0 && params_stride != 6) return ta_fail(TA_EINVAL, "params_stride must be 0 or 6");
    if (score_bound < 0 || score_bound >= (1ll << 23))
        return ta_fail(TA_ERANGE, "(n+m+2)*max|param| does not fit the 32-bit encoded scores");
    if (max_m > ta_nw2_max_m()) return ta_fail(TA_ELIMIT, "m exceeds the LDS capacity for the OCR codes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NwArgs a{t_codes, t_off, o_codes, o_off, params, params_stride, ws, ws_off,
             ops_out, ops_off, ops_len, nprob};
    if ((flags & TA_NW_FILL) && max_n > 0 && max_m > 0) {
        if (!t_codes || !o_codes || !ws) return ta_fail(TA_EINVAL, "null code/workspace pointer");
        if (flags & TA_NW_CHECK_IDS) {
            const int rc = check_ids(a, flags, st);
            if (rc != TA_OK) return rc;
        }
        const hipError_t e = launch_score(a, max_n, max_m, flags, st);
        if (e != hipSuccess) return ta_fail_hip(e, "nw_score_kernel launch");
    }
    if (flags & TA_NW_TRACEBACK) {
        if (!ops_out && (max_n + max_m) > 0) return ta_fail(TA_EINVAL, "null ops_out");
        // launch shape by batch size (ta_nw2_traceback_plan above): several waves per problem that speculate along the
        // path for small and medium batches, two problems per wave on half-strips for batches that fill the chip and
        // share one scoring system -- the two halves of a wave run in lockstep, and problems scored differently have
        // paths of very different length (the grid search, 2187 x 800 x 900 with a system per problem: 0.76 ms against
        // 0.59 with one wave per problem; the same shape under one system 0.52 against 0.53)
        // every length starts as -1: the multi-wave tracebacks wait for each other's tokens with BOUNDED spins, and a walk
        // that gave up must not leave the length of an earlier run (or of nothing) behind -- the host refuses a negative one
        {
            const hipError_t em = hipMemsetAsync(ops_len, 0xFF, sizeof(int32_t) * (size_t)nprob, st);
            if (em != hipSuccess) return ta_fail_hip(em, "ops_len reset");
        }
        const int tbw = ta_nw2_traceback_plan(nprob, params_stride, flags);
        if (tbw == 3) hipLaunchKernelGGL(nw_trace2h_kernel, dim3((nprob + 1) / 2), dim3(64), 0, st, a);
        else if (tbw == 5) hipLaunchKernelGGL(nw_trace2hw_kernel<2>, dim3((nprob + 1) / 2), dim3(128), 0, st, a);
        else if (tbw == 6) hipLaunchKernelGGL(nw_trace2hw_kernel<4>, dim3((nprob + 1) / 2), dim3(256), 0, st, a);
        else if (tbw == 4) hipLaunchKernelGGL(nw_trace2w_kernel<4>, dim3(nprob), dim3(256), 0, st, a);
        else if (tbw == 2) hipLaunchKernelGGL(nw_trace2w_kernel<2>, dim3(nprob), dim3(128), 0, st, a);
        else hipLaunchKernelGGL(nw_trace2_kernel, dim3(nprob), dim3(64), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ta_fail_hip(e, "nw_trace2_kernel launch");
    }
    return TA_OK;
}
