/* abi_c_example.c -- the C ABI of include/text_alignment_amd.h driven from plain C: no Python,
 * no torch, device memory from hipMalloc.  Aligns the reference's demo pair
 * (textSeqCompare.py:180-190 style: two short strings, default scoring) and prints the two
 * aligned strings; tests/test_abi.py builds this with gcc, runs it on the GPU box and compares
 * the output with the oracle.
 *
 *   gcc -std=c99 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ abi_c_example.c \
 *       -L text_alignment_amd -lta_hip -L /opt/rocm/lib -lamdhip64 -o abi_c_example
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "text_alignment_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static void* to_device(const void* src, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    const char* t = argc > 1 ? argv[1] : "dominus deus meus";
    const char* o = argc > 2 ? argv[2] : "domnus dcus  meuss";
    const int32_t n = (int32_t)strlen(t), m = (int32_t)strlen(o);
    int32_t params[6] = {8, -4, -7, -7, -3, 0};              /* textSeqCompare.py:10 */
    int32_t* tc = malloc(sizeof(int32_t) * (n + 1));
    int32_t* oc = malloc(sizeof(int32_t) * (m + 1));
    for (int i = 0; i < n; ++i) tc[i] = (unsigned char)t[i];   /* equal ids <=> equal tokens */
    for (int j = 0; j < m; ++j) oc[j] = (unsigned char)o[j];
    int64_t t_off[2] = {0, n}, o_off[2] = {0, m}, zero = 0;

    const int64_t ws_bytes = ta_nw_workspace_bytes(n, m);
    void *d_t = to_device(tc, sizeof(int32_t) * n), *d_o = to_device(oc, sizeof(int32_t) * m);
    void *d_toff = to_device(t_off, sizeof t_off), *d_ooff = to_device(o_off, sizeof o_off);
    void *d_prm = to_device(params, sizeof params), *d_zero = to_device(&zero, sizeof zero);
    void *d_ws = to_device(NULL, 0), *d_ops = NULL, *d_len = NULL;
    if (ws_bytes > 0) { CHECK_HIP(hipFree(d_ws)); CHECK_HIP(hipMalloc(&d_ws, (size_t)ws_bytes)); }
    CHECK_HIP(hipMalloc(&d_ops, (size_t)(n + m + 16)));
    CHECK_HIP(hipMalloc(&d_len, sizeof(int32_t)));
    if (!d_t || !d_o || !d_toff || !d_ooff || !d_prm || !d_zero) { fprintf(stderr, "alloc failed\n"); return 2; }

    const int64_t bound = (int64_t)(n + m + 2) * (3 * 8 + 2);
    int rc = ta_nw_batch(d_t, d_toff, d_o, d_ooff, 1, d_prm, 0, d_ws, d_zero, d_ops, d_zero, d_len,
                         n, m, bound, TA_NW_FILL | TA_NW_TRACEBACK, NULL /* default stream */);
    if (rc != 0) { fprintf(stderr, "ta_nw_batch: %d %s\n", rc, ta_last_error()); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    int32_t len = 0;
    unsigned char* ops = malloc((size_t)(n + m + 16));
    CHECK_HIP(hipMemcpy(&len, d_len, sizeof len, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ops, d_ops, (size_t)(n + m), hipMemcpyDeviceToHost));

    /* columns are right-aligned in the caller's region: 0 pair, 1 (t,'_'), 2 ('_',o) */
    char* ta = malloc((size_t)len + 1); char* oa = malloc((size_t)len + 1);
    int i = 0, j = 0;
    for (int k = 0; k < len; ++k) {
        const unsigned char c = ops[n + m - len + k];
        ta[k] = (c != 2) ? t[i++] : '_';
        oa[k] = (c != 1) ? o[j++] : '_';
    }
    ta[len] = oa[len] = 0;
    printf("%s\n%s\n", ta, oa);
    if (!(i == n && j == m)) return 3;

    /* The two-phase aligner and its debug guard (TA_NW_CHECK_IDS).  The ids here are character codes (< 128):
     * asserting an alphabet of 27 is a caller bug that would silently give wrong alignments -- with the guard the
     * call is refused (TA_EINVAL) before anything is launched; asserting 128 is true, and the call gives the
     * alignment of the one-pass kernel above. */
    const int64_t ws2_bytes = ta_nw2_workspace_bytes(n, m);
    void* d_ws2 = NULL;
    CHECK_HIP(hipMalloc(&d_ws2, (size_t)(ws2_bytes > 16 ? ws2_bytes : 16)));
    CHECK_HIP(hipMemset(d_len, 0, sizeof(int32_t)));
    const uint32_t base = TA_NW_FILL | TA_NW_TRACEBACK | TA_NW_CHECK_IDS;
    const int rc_bad = ta_nw2_batch(d_t, d_toff, d_o, d_ooff, 1, d_prm, 0, d_ws2, d_zero, d_ops, d_zero, d_len,
                                    n, m, bound, base | TA_NW_CODES8 | TA_NW_ALPHABET(27), NULL);
    const int rc_ok = ta_nw2_batch(d_t, d_toff, d_o, d_ooff, 1, d_prm, 0, d_ws2, d_zero, d_ops, d_zero, d_len,
                                   n, m, bound, base | TA_NW_CODES8 | TA_NW_ALPHABET(128), NULL);
    CHECK_HIP(hipDeviceSynchronize());
    int32_t len2 = 0;
    unsigned char* ops2 = malloc((size_t)(n + m + 16));
    CHECK_HIP(hipMemcpy(&len2, d_len, sizeof len2, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ops2, d_ops, (size_t)(n + m), hipMemcpyDeviceToHost));
    const int same = len2 == len && memcmp(ops2 + n + m - len2, ops + n + m - len, (size_t)len) == 0;
    printf("check_ids: wrong alphabet rc=%d, true alphabet rc=%d, two-phase equals one-pass=%d\n", rc_bad, rc_ok, same);
    return (rc_bad == TA_EINVAL && rc_ok == TA_OK && same) ? 0 : 4;
}
