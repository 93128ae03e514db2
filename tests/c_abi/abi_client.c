/* abi_client.c -- a plain-C client of the engine's C ABI (no Python, no torch in the process).
 *
 *   abi_client <weights.bin> <ids.bin> <B> <L> <precision: 0 fp32 | 1 bf16 | 2 fp16 | 3 fp16c | 4 fp16x3>
 *
 * weights.bin: repeated records  { u32 key_len, key bytes, u32 ndim, i64 shape[ndim], f32 data[prod(shape)] }  (written by
 * tests/test_gpu_c_abi.py from a state_dict with the reference's checkpoint keys); ids.bin: B*L token ids as uint8.
 * Prints one line per read: "<logit0> <logit1>" (%.9g).  Every call goes through include/chimeralm_hip.h exactly as a
 * foreign-language binding would; device memory comes from the HIP runtime's C API.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "chimeralm_hip.h"

#define CHECK_HIP(x)                                                                 \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            return 2;                                                                \
        }                                                                            \
    } while (0)
#define CHECK_CLM(h, x)                                                              \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != CLM_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, clm_last_error(h));           \
            return 3;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s weights.bin ids.bin B L precision\n", argv[0]);
        return 1;
    }
    const int B = atoi(argv[3]), L = atoi(argv[4]), prec = atoi(argv[5]);
    if (clm_abi_version() != CLM_ABI_VERSION) {
        fprintf(stderr, "ABI version mismatch\n");
        return 1;
    }
    clm_config cfg;
    clm_handle* h = NULL;
    CHECK_CLM(h, clm_default_config(&cfg));
    cfg.precision = prec;
    cfg.chunk_reads = 4;
    CHECK_CLM(h, clm_create(&cfg, 0, &h));

    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    for (;;) {
        uint32_t klen, ndim;
        if (fread(&klen, 4, 1, f) != 1) break;
        char key[512];
        int64_t shape[8];
        if (klen >= sizeof key || fread(key, 1, klen, f) != klen || fread(&ndim, 4, 1, f) != 1 || ndim > 8 ||
            fread(shape, 8, ndim, f) != ndim) { fprintf(stderr, "corrupt weights file\n"); return 1; }
        key[klen] = 0;
        size_t n = 1;
        for (uint32_t i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        float* data = (float*)malloc(n * sizeof(float));
        if (!data || fread(data, sizeof(float), n, f) != n) { fprintf(stderr, "short read for %s\n", key); return 1; }
        CHECK_CLM(h, clm_load_weight(h, key, data, CLM_DT_F32, shape, (int)ndim));   /* host pointer: copied by the engine */
        free(data);
    }
    fclose(f);
    CHECK_CLM(h, clm_finalize(h));

    uint8_t* ids = (uint8_t*)malloc((size_t)B * L);
    f = fopen(argv[2], "rb");
    if (!f || fread(ids, 1, (size_t)B * L, f) != (size_t)B * L) { fprintf(stderr, "cannot read ids\n"); return 1; }
    fclose(f);
    void* d_ids = NULL;
    float* d_logits = NULL;
    hipStream_t stream;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc(&d_ids, (size_t)B * L));
    CHECK_HIP(hipMalloc((void**)&d_logits, (size_t)B * 2 * sizeof(float)));
    CHECK_HIP(hipMemcpyAsync(d_ids, ids, (size_t)B * L, hipMemcpyHostToDevice, stream));
    CHECK_CLM(h, clm_reserve(h, B, L));
    CHECK_CLM(h, clm_forward(h, d_ids, CLM_DT_U8, L, B, L, d_logits, stream));       /* asynchronous on `stream` */
    float* logits = (float*)malloc((size_t)B * 2 * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(logits, d_logits, (size_t)B * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    for (int b = 0; b < B; ++b) printf("%.9g %.9g\n", logits[2 * b], logits[2 * b + 1]);

    /* the mode against the exact-fp32 kernels of the same handle, on these reads (stderr: "selfcheck <max |dlogit|> <labels differing>") */
    {
        float diff = -1.f;
        int differ = -1;
        CHECK_CLM(h, clm_selfcheck(h, d_ids, CLM_DT_U8, L, B, L, stream, &diff, &differ));
        fprintf(stderr, "selfcheck %.9g %d\n", diff, differ);
        if (clm_effective_precision(h, L) < 0) return 5;
    }

    /* error convention: a bad call returns a negative code and leaves a message, nothing is thrown */
    if (clm_forward(h, d_ids, CLM_DT_U8, L, -1, L, d_logits, stream) != CLM_E_INVALID || !clm_last_error(h)[0]) {
        fprintf(stderr, "expected CLM_E_INVALID for B = -1\n");
        return 4;
    }
    CHECK_HIP(hipFree(d_ids));
    CHECK_HIP(hipFree(d_logits));
    CHECK_HIP(hipStreamDestroy(stream));
    CHECK_CLM(h, clm_destroy(h));
    free(ids);
    free(logits);
    return 0;
}
