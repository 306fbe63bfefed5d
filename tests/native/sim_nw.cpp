// tests/native/sim_nw.cpp -- host-side lane simulator of the NW wavefront kernel (TEST ONLY).
//
// Replays the data flow of text_alignment_amd/csrc/ta_nw.hip on the CPU -- 64 lanes, R rows
// per lane, skewed steps, the wave_shr hand-down of V/D between lanes, the strip-to-strip
// hand-off row, the grouped 16-byte pointer stores and the traceback's addressing -- using
// the SAME nw_cell.h the kernel compiles, so the encoding, boundary formulas and layout are
// checked against the oracle without a GPU.  Build: g++ -O2 -shared -fPIC (tests do it).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../text_alignment_amd/csrc/nw_cell.h"

using namespace ta;

template <int R>
static int run(const int32_t* t, int n, const int32_t* o, int m, const int* p,
               uint8_t* ops_out, int* ops_len) {
    using L = PtrLayout<R>;
    const CellConsts c = make_consts(p[0], p[1], p[2], p[3], p[4], p[5]);
    std::vector<uint8_t> ptr((size_t)L::total_bytes(n, m) + 16, 0xEE);   // 0xEE = "never written"
    // hand-off row: hv[j], hd[j] = V, D emitted by the row above the current strip
    std::vector<int> hv(m + 2), hd(m + 2);
    for (int j = 0; j <= m; ++j) { hv[j] = bnd_V_row0(c, j); hd[j] = bnd_D_row0(c, j); }

    const int nsteps = L::nsteps(m), ngroups = L::ngroups(m);
    for (int s = 0; s < L::nstrips(n); ++s) {
        int D[kLanes][R], V[kLanes][R], H[kLanes][R], tcode[kLanes][R];
        int dsave[kLanes];
        uint8_t acc[kLanes][16];
        for (int l = 0; l < kLanes; ++l) {
            for (int r = 0; r < R; ++r) {
                const int i = s * L::SR + l * R + r + 1;
                D[l][r] = bnd_D_col0(c, i);
                H[l][r] = bnd_H_col0(c, i);
                V[l][r] = 0;
                tcode[l][r] = (i <= n) ? t[i - 1] : -1;
            }
            dsave[l] = bnd_D_col0(c, s * L::SR + l * R);
        }
        for (int g = 0; g < ngroups; ++g) {
            for (int q = 0; q < L::SPG; ++q) {
                const int k = g * L::SPG + q;
                // --- cross-lane phase (full EXEC in the kernel) ---
                int vup[kLanes], dul0[kLanes], dsave_new[kLanes];
                for (int l = 0; l < kLanes; ++l) {
                    const int j = k - l + 1;
                    if (l == 0) {
                        const int jj = (j >= 1 && j <= m) ? j : 0;      // lane 0 reads hand-off row
                        vup[l] = hv[jj];
                        dsave_new[l] = hd[jj];
                    } else {
                        vup[l] = V[l - 1][R - 1];                        // wave_shr:1
                        dsave_new[l] = D[l - 1][R - 1];
                    }
                    dul0[l] = dsave[l];
                }
                // --- compute phase (EXEC = active lanes) ---
                int newD[kLanes][R], newV[kLanes][R], newH[kLanes][R];
                bool active[kLanes];
                for (int l = 0; l < kLanes; ++l) {
                    const int j = k - l + 1;
                    active[l] = (j >= 1 && j <= m);
                    if (!active[l]) continue;
                    int d_ul = dul0[l], v_u = vup[l];
                    for (int r = 0; r < R; ++r) {
                        const int cs = (tcode[l][r] == o[j - 1]) ? c.cmatch : c.cmismatch;
                        int d, v, h;
                        const unsigned b = cell_update(d_ul, v_u, H[l][r], cs, c.gox6, c.goy6, d, v, h);
                        acc[l][q * R + r] = (uint8_t)b;
                        d_ul = D[l][r];           // old D of this row = up-left of the next row
                        v_u = v;
                        newD[l][r] = d; newV[l][r] = v; newH[l][r] = h;
                    }
                }
                for (int l = 0; l < kLanes; ++l) {
                    if (!active[l]) continue;
                    for (int r = 0; r < R; ++r) { D[l][r] = newD[l][r]; V[l][r] = newV[l][r]; H[l][r] = newH[l][r]; }
                    dsave[l] = dsave_new[l];
                    if (l == kLanes - 1) {                               // lane 63 publishes its bottom row
                        const int j = k - l + 1;
                        hv[j] = V[l][R - 1];
                        hd[j] = D[l][R - 1];
                    }
                }
            }
            for (int l = 0; l < kLanes; ++l)
                memcpy(&ptr[(size_t)s * L::strip_bytes(m) + ((size_t)g * 64 + l) * 16], acc[l], 16);
        }
        (void)nsteps;
    }
    // traceback (textSeqCompare.py:96-164) through the layout
    int x = n, y = m, len = 0;
    std::vector<uint8_t> rev;
    int st = 0;
    if (n > 0 && m > 0) st = ptr_pm(ptr[(size_t)L::addr(n, m, m)]);
    while (x > 0 && y > 0) {
        const unsigned b = ptr[(size_t)L::addr(x, y, m)];
        if (b == 0xEE) return -7;   // would mean the walk read a byte the fill never wrote (0xEE is not a valid code: field 3)
        if (st == 0) { rev.push_back(0); st = ptr_pm(b); --x; --y; }
        else if (st == 1) { rev.push_back(1); st = ptr_px(b); --x; }
        else { rev.push_back(2); st = ptr_py(b); --y; }
    }
    while (y > 0) { rev.push_back(2); --y; }
    while (x > 0) { rev.push_back(1); --x; }
    len = (int)rev.size();
    for (int a = 0; a < len; ++a) ops_out[a] = rev[len - 1 - a];
    *ops_len = len;
    return 0;
}

extern "C" int sim_nw(const int32_t* t, int n, const int32_t* o, int m, const int* params, int R,
                      uint8_t* ops_out, int* ops_len) {
    switch (R) {
        case 4: return run<4>(t, n, o, m, params, ops_out, ops_len);
        case 8: return run<8>(t, n, o, m, params, ops_out, ops_len);
        case 16: return run<16>(t, n, o, m, params, ops_out, ops_len);
        default: return -1;
    }
}

// raw pointer bytes in reference order, for a cell-by-cell comparison with the oracle
extern "C" int sim_layout_addr(int R, int i, int j, int m, int64_t* out) {
    switch (R) {
        case 4: *out = PtrLayout<4>::addr(i, j, m); return 0;
        case 8: *out = PtrLayout<8>::addr(i, j, m); return 0;
        case 16: *out = PtrLayout<16>::addr(i, j, m); return 0;
        default: return -1;
    }
}
