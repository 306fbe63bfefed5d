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
#include <algorithm>
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

// ---------------------------------------------------------------------------------------------
// Two-phase aligner: phase 1 fills scores only (raw integers, no pointer bytes) and checkpoints
// (a) every KCG groups the whole lane state of the strip's wave and (b) per strip the bottom row
// (XG or V~, D) that the strip below starts from; phase 2 walks back strip by strip, re-running the
// TAGGED fill only over a window of skewed steps [g0*SPG, k_in] restarted from a checkpoint (two
// halo steps make the winner tags of the restart state irrelevant) and walking the pointer bytes
// of that window.  The stored bottom rows carry no winner tags, so the two pointers a strip's first
// row takes from the row above (PM, PX) are not in the window: a step that leaves the strip upwards
// is taken with its next state PENDING, and the state is read off the tagged outputs of the cell it
// lands on when the strip above is re-filled.
template <int R>
static int run2(const int32_t* t, int n, const int32_t* o, int m, const int* p, int KCG, int GSPAN,
                uint8_t* ops_out, int* ops_len) {
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    const CellConsts c = make_consts(p[0], p[1], p[2], p[3], p[4], p[5]);
    const int cmat_raw = p[0] - p[4] - p[5], cmis_raw = p[1] - p[4] - p[5];
    const int nstrips = L::nstrips(n), ngroups = L::ngroups(m);
    const int nck = ngroups / KCG + 1;
    struct State { int D[kLanes][R], H[kLanes][R], Vlast[kLanes], dsave[kLanes]; };
    std::vector<State> ck_((size_t)std::max(nstrips, 1) * nck);
    // bottom rows: row s is what strip s starts from (row 0 = the table's boundary row), strip s
    // leaves row s + 1; index [row][j], j = 0..m
    std::vector<int> HV((size_t)(std::max(nstrips, 1) + 1) * (m + 2)), HD(HV.size());
    // ... and, for the half-strip flow of nw_trace2h_kernel, the bottom rows of lanes 31 and 63 of every strip
    // (Ws2 with kSubRows = 2): sub row 0 = the table's boundary row, sub row 1 + 2 s + q = last row of lanes
    // 32 q .. 32 q + 31 of strip s; the row above half-strip hs is sub row hs
    std::vector<int> SV((size_t)(2 * std::max(nstrips, 1) + 1) * (m + 2)), SD(SV.size());

    // phase 1 keeps V~ + gox / H~ + goy (the carried cell of nw_cell.h) when no gap open is
    // positive, exactly as nw_score_kernel does; phase 2 undoes the offsets where it reads them
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int xadj = carried ? c.gox : 0, yadj = carried ? c.goy : 0;

    // ---------------- phase 1: raw fill ----------------
    {
        for (int j = 0; j <= m; ++j) { HV[j] = raw_of(bnd_V_row0(c, j)) + xadj; HD[j] = raw_of(bnd_D_row0(c, j)); SV[j] = HV[j]; SD[j] = HD[j]; }
        for (int s = 0; s < nstrips; ++s) {
            const int* hv = &HV[(size_t)s * (m + 2)];
            const int* hd = &HD[(size_t)s * (m + 2)];
            int* hv_out = &HV[(size_t)(s + 1) * (m + 2)];
            int* hd_out = &HD[(size_t)(s + 1) * (m + 2)];
            int D[kLanes][R], V[kLanes][R], H[kLanes][R], tcode[kLanes][R], dsave[kLanes];
            for (int l = 0; l < kLanes; ++l) {
                for (int r = 0; r < R; ++r) {
                    const int i = s * L::SR + l * R + r + 1;
                    D[l][r] = raw_of(bnd_D_col0(c, i)); H[l][r] = raw_of(bnd_H_col0(c, i)) + yadj; V[l][r] = 0;
                    tcode[l][r] = (i <= n) ? t[i - 1] : -1;
                }
                dsave[l] = raw_of(bnd_D_col0(c, s * L::SR + l * R));
            }
            for (int g = 0; g < ngroups; ++g) {
                if (g > 0 && g % KCG == 0) {
                    State& st = ck_[(size_t)s * nck + g / KCG];
                    for (int l = 0; l < kLanes; ++l) {
                        for (int r = 0; r < R; ++r) { st.D[l][r] = D[l][r]; st.H[l][r] = H[l][r]; }
                        st.Vlast[l] = V[l][R - 1]; st.dsave[l] = dsave[l];
                    }
                }
                for (int q = 0; q < SPG; ++q) {
                    const int k = g * SPG + q;
                    int vup[kLanes], dnext[kLanes];
                    for (int l = 0; l < kLanes; ++l) {
                        const int j = k - l + 1;
                        if (l == 0) { const int jj = (j >= 1 && j <= m) ? j : 0; vup[l] = hv[jj]; dnext[l] = hd[jj]; }
                        else { vup[l] = V[l - 1][R - 1]; dnext[l] = D[l - 1][R - 1]; }
                    }
                    int nD[kLanes][R], nV[kLanes][R], nH[kLanes][R];
                    bool act[kLanes];
                    for (int l = 0; l < kLanes; ++l) {
                        const int j = k - l + 1;
                        act[l] = (j >= 1 && j <= m);
                        if (!act[l]) continue;
                        int d_ul = dsave[l], v_u = vup[l];
                        for (int r = 0; r < R; ++r) {
                            const int cs = (tcode[l][r] == o[j - 1]) ? cmat_raw : cmis_raw;
                            int d, v, h;
                            if (carried) cell_update_carried(d_ul, v_u, H[l][r], cs, c.gox, c.goy, d, v, h);
                            else cell_update_raw(d_ul, v_u, H[l][r], cs, c.gox, c.goy, d, v, h);
                            d_ul = D[l][r]; v_u = v;
                            nD[l][r] = d; nV[l][r] = v; nH[l][r] = h;
                        }
                    }
                    for (int l = 0; l < kLanes; ++l) {
                        if (!act[l]) continue;
                        for (int r = 0; r < R; ++r) { D[l][r] = nD[l][r]; V[l][r] = nV[l][r]; H[l][r] = nH[l][r]; }
                        dsave[l] = dnext[l];
                        if (l == kLanes - 1) {
                            const int j = k - l + 1;
                            hv_out[j] = V[l][R - 1]; hd_out[j] = D[l][R - 1];
                        }
                        if ((l & 31) == 31) {
                            const int j = k - l + 1;
                            const size_t b = (size_t)(1 + 2 * s + l / 32) * (m + 2);
                            SV[b + j] = V[l][R - 1]; SD[b + j] = D[l][R - 1];
                        }
                    }
                }
            }
        }
    }

    // ---------------- phase 2: chunked tagged re-fill + walk (nw_trace2_kernel's data flow) ----------------
    // A chunk = the KCG groups between two state checkpoints.  The chunk the walk is in is re-filled
    // with the tagged cell from its checkpoint (in carried form when phase 1 used it); its first two
    // steps carry no valid tags, so when the walk gets there the chunk before is re-filled one group
    // further (g_top = the first group of the chunk just left).  Of a chunk's pointer bytes only the lanes
    // l_lo .. l_lo + WL - 1 are kept, l_lo = max(0, entry lane - WL + 1) (WL = GSPAN if 0 < GSPAN < 64, else all
    // 64: nw_trace2_kernel's kWinLanes): a walk that needs a lane above l_lo stops, and the SAME chunk is
    // re-filled up to the group the walk stands in with the window at its lane.  Bytes outside the window
    // stay 0xEE here, so a walk that read one would fail with -7.
    // GSPAN >= 1000: the data flow of nw_trace2w_kernel (several waves per problem, each re-filling the chunk it
    // expects into a buffer of its own): whole chunks are kept (WL = 64; the entry lane of a chunk re-filled ahead of
    // the walk is not known), a chunk entered through its halo is re-filled to the first group of the chunk just left
    // with two steps of it -- what the one-wave flow does as well, so an expected chunk that turns out right IS the
    // chunk the walk needs -- and a position in a chunk's first two steps goes to the chunk before at once (tb_job_at)
    // instead of through a walk of zero steps.
    // GSPAN >= 2000: the half-strip flow of nw_trace2h_kernel on top of that: a strip restarts in two halves of 32 lanes
    // (from the same state checkpoints and from lane 31's bottom row), a step that leaves a HALF-strip upwards is
    // pending, and a half-strip's whole chunk is kept.
    const bool halves = GSPAN >= 2000;
    if (halves) GSPAN -= 1000;
    const bool skip_halo = GSPAN >= 1000;
    if (skip_halo) GSPAN -= 1000;
    const int LWs = halves ? 32 : kLanes;                   // lanes of the unit phase 2 restarts
    const int xadj6 = xadj * 64, yadj6 = yadj * 64;
    std::vector<uint8_t> rev;
    int x = n, y = m, st = 0;
    bool first = true;
    // pend: 0, or 3 + (state the strip was left in): the next state is the winner tag of D (3, left
    // in state M) or of XG / V~ (4, left in state X) of the cell (x, y) the walk now stands on.
    int pend = 0;
    bool probe = false;                               // start state of a walk that begins in a strip's first row
    if ((n - 1) % (LWs * R) == 0 && n > 1) {          // textSeqCompare.py:102 reads PM(n, m) = tag of D(n-1, m-1)
        if (m == 1) { st = 0; first = false; }        // D(n-1, 0): boundary column, M
        else { x = n - 1; y = m - 1; pend = 3; probe = true; }
    }
    int guard = 0;
    while (x > 0 && y > 0) {
        const int s = (x - 1) / L::SR;
        const int hs = (x - 1) / (LWs * R);                  // (half-)strip index; lanes lb .. lb + LWs - 1 of strip s
        const int lb = halves ? (hs & 1) * LWs : 0;
        int l = ((x - 1) % L::SR) / R, r = (x - 1) % R, k = (y - 1) + l;
        int ck = (k / SPG) / KCG;
        int g_top = k / SPG;
        if (skip_halo && ck > 0 && k < ck * KCG * SPG + 2) ck -= 1;      // tb_job_at
        // a chunk re-filled AHEAD of the walk for a strip entry (tb_predict) goes two groups further than the diagonal's
        // entry group, whole groups: a window re-filled further than the walk's entry point serves it as well
        if (skip_halo) g_top = std::min(g_top + 2, ck * KCG + KCG);     // (nw_trace2w_kernel and, on half-strips, nw_trace2hw_kernel)
        bool in_strip = true;
        while (in_strip) {
            if (++guard > 8 * (n + m) + 64) return -9;
            const int g0 = ck * KCG, k0 = g0 * SPG;
            const int kvalid = ck > 0 ? k0 + 2 : 0;
            // the row above the strip for columns k0 .. min(m, (g_top+1)*SPG), x-input in the re-fill's
            // form; tags only where they are known analytically (the table's boundary row)
            const int jhi = std::min(m, (g_top + 1) * SPG);
            std::vector<int> hvt(m + 2, 0), hdt(m + 2, 0);
            for (int j = std::max(0, k0 - lb); j <= jhi; ++j) {
                if (hs == 0) { hvt[j] = bnd_V_row0(c, j) + xadj6; hdt[j] = bnd_D_row0(c, j); continue; }
                if (halves) {
                    const size_t b = (size_t)hs * (m + 2);
                    hvt[j] = enc_of(SV[b + j]); hdt[j] = enc_of(SD[b + j]);
                    if (j == 0) { hvt[j] = 0; hdt[j] = bnd_D_col0(c, hs * LWs * R); }   // column 0 of the row above (kernel: analytic)
                    continue;
                }
                const size_t b = (size_t)s * (m + 2);
                hvt[j] = enc_of(HV[b + j]); hdt[j] = enc_of(HD[b + j]);
            }
            // lane state at the start of group g0
            int D[kLanes][R], V[kLanes][R], H[kLanes][R], tcode[kLanes][R], dsave[kLanes];
            for (int ll = 0; ll < kLanes; ++ll) {
                for (int rr = 0; rr < R; ++rr) {
                    const int i = s * L::SR + ll * R + rr + 1;
                    tcode[ll][rr] = (i <= n) ? t[i - 1] : -1;
                    V[ll][rr] = 0;
                    D[ll][rr] = bnd_D_col0(c, i); H[ll][rr] = bnd_H_col0(c, i) + yadj6;
                }
                dsave[ll] = bnd_D_col0(c, s * L::SR + ll * R);
            }
            if (g0 > 0) {
                // lanes that have not started by step k0 (lane >= k0: only with checkpoint periods
                // shorter than 64 steps) keep the TAGGED boundary values: the scores are the same
                // and their column-0 tags are pointers of the column-1 cells
                const State& cs0 = ck_[(size_t)s * nck + g0 / KCG];
                for (int ll = 0; ll < std::min(kLanes, k0); ++ll) {
                    for (int rr = 0; rr < R; ++rr) { D[ll][rr] = enc_of(cs0.D[ll][rr]); H[ll][rr] = enc_of(cs0.H[ll][rr]); }
                    V[ll][R - 1] = enc_of(cs0.Vlast[ll]); dsave[ll] = enc_of(cs0.dsave[ll]);
                }
            }
            // tagged fill of groups g0..g_top into the chunk buffer; the tagged outputs of the strip's
            // bottom row are kept per step for a pending state
            const int WL = halves ? LWs : (GSPAN > 0 && GSPAN < 64) ? GSPAN : 64;
            const int l_lo = halves ? lb : std::max(0, l - (WL - 1));
            std::vector<uint8_t> wbuf((size_t)(g_top - g0 + 1) * 1024, 0xEE);
            std::vector<int> capV((size_t)(g_top - g0 + 1) * SPG, 0), capD(capV.size(), 0);
            std::vector<char> capOk(capV.size(), 0);
            for (int g = g0; g <= g_top; ++g) {
                uint8_t acc[kLanes][16];
                memset(acc, 0xEE, sizeof(acc));
                // of the group the walk enters the chunk in only the steps up to the entry point are re-filled (two
                // of four when it stands in the group's first half, as the kernel does): the rest stays poisoned
                const int nq = (g == g_top && k / SPG == g_top && (k % SPG) + 1 <= 2) ? 2 : SPG;
                for (int q = 0; q < nq; ++q) {
                    const int kk = g * SPG + q;
                    int vup[kLanes], dnext[kLanes];
                    for (int ll = lb; ll < lb + LWs; ++ll) {
                        const int j = kk - ll + 1;
                        if (ll == lb) { const int jj = std::min(std::max(j, 0), m); vup[ll] = hvt[jj]; dnext[ll] = hdt[jj]; }
                        else { vup[ll] = V[ll - 1][R - 1]; dnext[ll] = D[ll - 1][R - 1]; }
                    }
                    int nD[kLanes][R], nV[kLanes][R], nH[kLanes][R];
                    bool act[kLanes] = {false};
                    for (int ll = lb; ll < lb + LWs; ++ll) {
                        const int j = kk - ll + 1;
                        act[ll] = (j >= 1 && j <= m);
                        if (!act[ll]) continue;
                        int d_ul = dsave[ll], v_u = vup[ll];
                        for (int rr = 0; rr < R; ++rr) {
                            const int cs = (tcode[ll][rr] == o[j - 1]) ? c.cmatch : c.cmismatch;
                            int d, v, h;
                            const unsigned b = carried
                                ? cell_update_carried_tagged(d_ul, v_u, H[ll][rr], cs, c.gox6, c.goy6, d, v, h)
                                : cell_update(d_ul, v_u, H[ll][rr], cs, c.gox6, c.goy6, d, v, h);
                            acc[ll][q * R + rr] = (uint8_t)(b & 0x3F);
                            d_ul = D[ll][rr]; v_u = v;
                            nD[ll][rr] = d; nV[ll][rr] = v; nH[ll][rr] = h;
                        }
                    }
                    for (int ll = lb; ll < lb + LWs; ++ll) {
                        if (!act[ll]) continue;
                        for (int rr = 0; rr < R; ++rr) { D[ll][rr] = nD[ll][rr]; V[ll][rr] = nV[ll][rr]; H[ll][rr] = nH[ll][rr]; }
                        dsave[ll] = dnext[ll];
                    }
                    if (act[lb + LWs - 1]) {                  // the (half-)strip's bottom row, tagged
                        capV[kk - k0] = V[lb + LWs - 1][R - 1]; capD[kk - k0] = D[lb + LWs - 1][R - 1]; capOk[kk - k0] = 1;
                    }
                }
                for (int ll = l_lo; ll < std::min(kLanes, l_lo + WL); ++ll) memcpy(&wbuf[((size_t)(g - g0) * 64 + ll) * 16], acc[ll], 16);
            }
            auto byte_at = [&](int ll, int rr, int kk) -> unsigned {
                return wbuf[((size_t)(kk / SPG - g0) * 64 + ll) * 16 + (kk % SPG) * R + rr];
            };
            if (pend) {                                   // the cell the walk stands on: bottom row of this strip
                if (l != lb + LWs - 1 || r != R - 1) return -10;
                if (k < k0 || k / SPG > g_top || !capOk[k - k0]) return -11;
                const int tagged = (pend == 3) ? capD[k - k0] : capV[k - k0];
                st = 2 - (tagged & 3);
                pend = 0;
                if (probe) {                              // that was the start state: back to (n, m)
                    probe = false; first = false;
                    x = n; y = m;
                    break;
                }
            }
            if (first && k >= kvalid) { st = ptr_pm(byte_at(l, r, k)); first = false; }   // start state, textSeqCompare.py:102
            int steps = 0;
            while (x > 0 && y > 0 && l >= l_lo && k >= kvalid) {
                if (k / SPG > g_top) return -6;               // would read past what this chunk re-filled
                const unsigned b = byte_at(l, r, k);
                if (b == 0xEE) return -7;
                const int up = (st != 2), left = (st != 1);
                rev.push_back((uint8_t)st);
                const bool leaves_up = up && l == lb && r == 0 && hs > 0;   // PM / PX of the (half-)strip's first row
                if (leaves_up) pend = 3 + st;
                else st = 2 - (int)((b >> (2 * st)) & 3u);
                const int wrap = up & (r == 0);
                r = (r - up) & (R - 1);
                x -= up; y -= left; k -= left + wrap; l -= wrap;
                ++steps;
            }
            if (x <= 0 || y <= 0 || l < lb) in_strip = false;
            else if (k >= kvalid) {
                if (steps == 0) return -12;                   // no progress: the entry lane is always in its window
                if (skip_halo) return -13;                    // whole chunks are kept: the walk cannot run off a window
                g_top = k / SPG;                              // ran off the window's top lane: same chunk, window at l
            } else {
                if (skip_halo && steps == 0) return -14;     // tb_job_at never starts a walk in a chunk's halo
                g_top = g0;
                ck -= 1;
                if (ck < 0) return -5;
            }
        }
    }
    while (y > 0) { rev.push_back(2); --y; }
    while (x > 0) { rev.push_back(1); --x; }
    const int len = (int)rev.size();
    for (int a = 0; a < len; ++a) ops_out[a] = rev[len - 1 - a];
    *ops_len = len;
    return 0;
}

extern "C" int sim_nw2(const int32_t* t, int n, const int32_t* o, int m, const int* params, int R,
                       int KCG, int GSPAN, uint8_t* ops_out, int* ops_len) {
    switch (R) {
        case 4: return run2<4>(t, n, o, m, params, KCG, GSPAN, ops_out, ops_len);
        case 8: return run2<8>(t, n, o, m, params, KCG, GSPAN, ops_out, ops_len);
        default: return -1;
    }
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
