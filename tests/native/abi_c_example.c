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
    return (i == n && j == m) ? 0 : 3;
}
