// nw_cell.h -- per-cell arithmetic and pointer-matrix layout of the NW wavefront kernel.
//
// Shared by the HIP kernels (ta_nw.hip) and by the host-side lane simulator used in the
// CPU tests (tests/native/sim_nw.cpp), so the encoding, boundary formulas and addressing
// are exercised without a GPU.
//
// What it computes: the affine-gap recurrence of the reference
// (textSeqCompare.py:62-88) with its boundary rows (textSeqCompare.py:53-60) and
// "first maximum wins" choice (list.index(max(..)), textSeqCompare.py:72,80,88).
//
// Formulation.  For each cell c = (i, j) the kernel keeps the three values the cell
// EMITS to its neighbours instead of M/X/Y themselves:
//     D(c) = max3(M, X, Y)                     -> M of (i+1, j+1)  (textSeqCompare.py:70-71)
//     V(c) = max3(M+gox+gex, X+gex, Y+gox+gex) -> X of (i+1, j)    (textSeqCompare.py:83-87)
//     H(c) = max3(M+goy+gey, X+goy+gey, Y+gey) -> Y of (i, j+1)    (textSeqCompare.py:75-79)
// and the winner index of each max3 is the pointer the receiving cell stores
// (PM, PX, PY).  Scores are kept relative to gex*i + gey*j ("hatted"), which removes the
// extension adds:  M^ = M - gex*i - gey*j etc.;  M^(i,j) = D^(i-1,j-1) + (s - gex - gey),
// X^(i,j) = V^(i-1,j), Y^(i,j) = H^(i,j-1), V^ = max3(M^+gox, X^, Y^+gox),
// H^ = max3(M^+goy, X^+goy, Y^).  The kernel further carries V~ = V^ - gox and
// H~ = H^ - goy (so X~ = X^ - gox, Y~ = Y^ - goy arrive ready-made), which leaves two adds
// per cell:  D = max3(M^, X~+gox, Y~+goy), V~ = max3(M^, X~, Y~+goy), H~ = max3(M^, X~+gox, Y~).
// A common offset inside one max3 never changes its winner.
//
// Encoding.  A value is stored as  (score << 6) | tag  with a static 6-bit tag per source
// matrix: M -> 0b101010, X -> 0b010101, Y -> 0.  A signed max over encoded candidates
// picks the larger score and, on equal scores, the larger tag = the earlier candidate in
// the reference's [from-M, from-X, from-Y] order, so one v_max3_i32 yields value AND
// winner.  The tag is replicated in three 2-bit fields so the pointer byte of a cell is
//     (D_ul & 0x03) | (V_u & 0x0C) | (H_l & 0x30)        (two v_bfi_b32)
// Field value f in {2,1,0} means pointer 2 - f in the reference's {0,1,2} numbering.
// Bits 6-7 of the stored byte are don't-care.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TA_HD __host__ __device__ __forceinline__
#else
#define TA_HD inline
#endif

namespace ta {

constexpr int kShift = 6;
constexpr int kTagM = 0x2A;
constexpr int kTagX = 0x15;
constexpr int kTagY = 0x00;
constexpr int kTagMask = 0x3F;
constexpr int kLanes = 64;

// Integer scoring parameters after the hatted transform, pre-shifted.
struct CellConsts {
    int cmatch;     // ((match    - gex - gey) << 6) | kTagM
    int cmismatch;  // ((mismatch - gex - gey) << 6) | kTagM
    int gox6;       // gox << 6
    int goy6;       // goy << 6
    int gox, goy, gex, gey;
};

TA_HD CellConsts make_consts(int match, int mismatch, int gox, int goy, int gex, int gey) {
    CellConsts c;
    c.cmatch = ((match - gex - gey) * 64) | kTagM;
    c.cmismatch = ((mismatch - gex - gey) * 64) | kTagM;
    c.gox6 = gox * 64;
    c.goy6 = goy * 64;
    c.gox = gox; c.goy = goy; c.gex = gex; c.gey = gey;
    return c;
}

TA_HD int max3i(int a, int b, int c) {
    int m = a > b ? a : b;
    return m > c ? m : c;
}

// ---- boundary cells (textSeqCompare.py:53-60; G = -1 always, textSeqCompare.py:9) ----
// Row 0: M = X = -j, Y = -inf.  Column 0: M = Y = -i, X = -inf.  At (0,0) M = X = 0.
// Hatted: row 0 -> -(1+gey)*j ; column 0 -> -(1+gex)*i.  The -inf entries never win
// (a finite candidate always exists), so they are resolved here analytically and the
// sentinel never enters the kernel.
TA_HD int bnd_D_row0(const CellConsts& c, int j) {            // D(0, j): M beats X on the tie
    return ((-(1 + c.gey) * j) * 64) | kTagM;
}
TA_HD int bnd_V_row0(const CellConsts& c, int j) {            // V~(0, j) = max(M^, X^ - gox)
    const int base = -(1 + c.gey) * j;
    return c.gox >= 0 ? ((base * 64) | kTagM) : (((base - c.gox) * 64) | kTagX);
}
TA_HD int bnd_D_col0(const CellConsts& c, int i) {            // D(i, 0): M beats Y on the tie
    return ((-(1 + c.gex) * i) * 64) | kTagM;
}
TA_HD int bnd_H_col0(const CellConsts& c, int i) {            // H~(i, 0) = max(M^, Y^ - goy)
    const int base = -(1 + c.gex) * i;
    return c.goy >= 0 ? ((base * 64) | kTagM) : (((base - c.goy) * 64) | kTagY);
}

// ---- one interior cell ----
// in : d_ul = D(i-1,j-1), v_u = V~(i-1,j), h_l = H~(i,j-1), cs = cmatch / cmismatch
// out: d, v, h = D, V~, H~ of (i,j); returns the pointer byte of (i,j) (bits 6-7 don't-care)
TA_HD unsigned cell_update(int d_ul, int v_u, int h_l, int cs, int gox6, int goy6,
                           int& d, int& v, int& h) {
    const int mr = (d_ul & ~kTagMask) + cs;            // M^ tagged M
    const int xr = (v_u & ~kTagMask) | kTagX;          // X~ tagged X
    const int yr = (h_l & ~kTagMask);                  // Y~ tagged Y (= 0)
    const int xg = xr + gox6;                          // X^
    const int yg = yr + goy6;                          // Y^
    d = max3i(mr, xg, yg);
    v = max3i(mr, xr, yg);
    h = max3i(mr, xg, yr);
    const unsigned inner = ((unsigned)v_u & 0x0Cu) | ((unsigned)h_l & ~0x0Cu);   // v_bfi_b32
    return ((unsigned)d_ul & 0x03u) | (inner & ~0x03u);                           // v_bfi_b32
}

// ---- score-only cell (no tags, no pointer byte): same three maxima on raw integers ----
// Used by the checkpointing fill (phase 1 of the two-phase aligner); the tagged cell above
// produces the same scores in its upper 26 bits: cell_update(...) >> 6 == cell_update_raw(...).
TA_HD void cell_update_raw(int d_ul, int v_u, int h_l, int cs, int gox, int goy,
                           int& d, int& v, int& h) {
    const int mr = d_ul + cs;
    const int xg = v_u + gox;
    const int yg = h_l + goy;
    d = max3i(mr, xg, yg);
    v = max3i(mr, v_u, yg);
    h = max3i(mr, xg, h_l);
}
// ---- score-only cell in CARRIED form, for non-positive gap opens (the reference's grid and its
// default system never open a gap for free, evaluate_text_alignment.py:181-188) ----
// State is XG = V~ + gox and YG = H~ + goy, i.e. what the max3 of the receiving cell consumes.
// With gox <= 0:  V~(c) + gox = max3(M^, X~, Y~+goy) + gox = max(D(c) + gox, XG_in)  because
// XG_in = X~ + gox >= X~ + 2 gox; likewise YG.  Same three scores as cell_update_raw, one add and
// one max3 fewer per cell:  D = max3(M^, XG, YG);  XG' = max(D + gox, XG);  YG' = max(D + goy, YG).
TA_HD bool opens_nonpositive(int gox, int goy) { return gox <= 0 && goy <= 0; }
TA_HD void cell_update_carried(int d_ul, int xg_u, int yg_l, int cs, int gox, int goy,
                               int& d, int& xg, int& yg) {
    const int mr = d_ul + cs;
    d = max3i(mr, xg_u, yg_l);
    const int dx = d + gox, dy = d + goy;
    xg = dx > xg_u ? dx : xg_u;
    yg = dy > yg_l ? dy : yg_l;
}
// ---- the carried cell on ENCODED values (score << 6 | tag): scores as cell_update_carried, and the
// winner tags of its three outputs are the pointers of the reference, like cell_update's.  Why the
// two 2-operand maxima pick the same winners as cell_update's max3 (a = M^, b = X~, c = Y~ + goy):
//   tag(D) = M: D = a;  XG' = max(a + gox [M], b + gox [X]) -> M iff a >= b            (= max3(a, b, c), a >= c)
//   tag(D) = X: D = b + gox <= b;  XG' = max(b + 2 gox [X], b + gox [X]) -> X            (a < b + gox <= b, b >= c)
//   tag(D) = Y: D = c;  XG' = max(c + gox [Y], b + gox [X]) -> X iff b >= c, else Y     (a < c)
// and symmetrically for YG'.  Needs gox, goy <= 0 (opens_nonpositive).
// in : d_ul = D(i-1,j-1), xg_u = XG(i-1,j), yg_l = YG(i,j-1) (any tags), cs = cmatch / cmismatch
// out: d, xg, yg of (i,j); returns the pointer byte of (i,j) (bits 6-7 don't-care)
TA_HD unsigned cell_update_carried_tagged(int d_ul, int xg_u, int yg_l, int cs, int gox6, int goy6,
                                          int& d, int& xg, int& yg) {
    const int mr = (d_ul & ~kTagMask) + cs;            // M^ tagged M
    const int xr = (xg_u & ~kTagMask) | kTagX;         // X~ + gox tagged X
    const int yr = (yg_l & ~kTagMask);                 // Y~ + goy tagged Y (= 0)
    d = max3i(mr, xr, yr);
    const int dx = d + gox6, dy = d + goy6;            // keep D's tag
    xg = dx > xr ? dx : xr;
    yg = dy > yr ? dy : yr;
    const unsigned inner = ((unsigned)xg_u & 0x0Cu) | ((unsigned)yg_l & ~0x0Cu);
    return ((unsigned)d_ul & 0x03u) | (inner & ~0x03u);
}
TA_HD int raw_of(int enc) { return enc >> kShift; }            // arithmetic shift: floor
TA_HD int enc_of(int raw) { return raw * 64; }                 // tag field zero

// ---- pointer-matrix layout (library-internal; the traceback kernel is its only reader) ----
// A strip is kLanes*R consecutive rows handled by one wave; lane l owns rows
// strip*SR + l*R + r (r < R).  The wave sweeps skewed steps k = (j-1) + l, so all lanes of a
// step lie on one anti-diagonal band.  Each lane collects R bytes per step and stores 16
// bytes every 16/R steps; the 64 lanes' 16-byte pieces of one store are contiguous (1 KiB,
// fully coalesced).  Byte address of cell (i, j), 1-based:
//     strip = (i-1) / SR, l = ((i-1) % SR) / R, r = (i-1) % R, k = (j-1) + l
//     group = k / SPG, q = k % SPG            (SPG = 16 / R steps per 16-byte piece)
//     addr  = strip * strip_bytes + (group * 64 + l) * 16 + q * R + r
template <int R>
struct PtrLayout {
    static constexpr int SR = kLanes * R;      // rows per strip
    static constexpr int SPG = 16 / R;         // steps per stored group
    TA_HD static int nsteps(int m) { return m + kLanes - 1; }
    TA_HD static int ngroups(int m) { return (nsteps(m) + SPG - 1) / SPG; }
    TA_HD static int nstrips(int n) { return (n + SR - 1) / SR; }
    TA_HD static int64_t strip_bytes(int m) { return (int64_t)ngroups(m) * 1024; }
    TA_HD static int64_t total_bytes(int n, int m) { return (int64_t)nstrips(n) * strip_bytes(m); }
    TA_HD static int64_t addr(int i, int j, int m) {
        const int i0 = i - 1;
        const int strip = i0 / SR;
        const int l = (i0 % SR) / R;
        const int r = i0 % R;
        const int k = (j - 1) + l;
        const int group = k / SPG, q = k % SPG;
        return (int64_t)strip * strip_bytes(m) + ((int64_t)group * 64 + l) * 16 + q * R + r;
    }
};

// pointer fields of a stored byte, in the reference's numbering (0 = from M, 1 = X, 2 = Y)
TA_HD int ptr_pm(unsigned b) { return 2 - (int)(b & 3u); }
TA_HD int ptr_px(unsigned b) { return 2 - (int)((b >> 2) & 3u); }
TA_HD int ptr_py(unsigned b) { return 2 - (int)((b >> 4) & 3u); }

}  // namespace ta
