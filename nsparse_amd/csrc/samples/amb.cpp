// amb_{s,d} <file.mtx> [seg_size block_size]  --  y = A x with the AMB format.
// Same command line, timing protocol and output lines as the reference driver
// (cuda-c/src/sample/spmv/spmv_amb.cu:15-118): CPU csr_kernel as the check, conversion timed
// with gettimeofday, 101 SpMV runs with the first discarded.  (Upstream tests argc >= 3 and
// then reads argv[3]; a manual plan needs both numbers, so this driver tests argc >= 4.)
#include <hip/hip_runtime.h>
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>

#include "nsparse.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s matrix.mtx [seg_size block_size]\n", argv[0]);
        return 1;
    }
    sfCSR mat;
    sfPlan plan;
    init_csr_matrix_from_file(&mat, argv[1]);
    real *x = (real *)malloc(sizeof(real) * mat.N);
    real *y = (real *)malloc(sizeof(real) * mat.M);
    init_vector(x, mat.N);
#ifdef sfDEBUG
    real *csr_y = (real *)malloc(sizeof(real) * mat.M);
    csr_kernel(csr_y, &mat, x);
#endif
    if (argc >= 4) set_plan(&plan, (size_t)atoi(argv[2]), atoi(argv[3]));
    else init_plan(&plan);
    // NSPARSE_BIN_CACHE=1: the plan found for this matrix is kept beside it
    const char *bc = getenv("NSPARSE_BIN_CACHE");
    const bool keep_plan = bc && bc[0] == '1' && argc < 4;
    char plan_path[4096];
    snprintf(plan_path, sizeof plan_path, "%s.plan", argv[1]);
    const bool had_plan = keep_plan && nsparse_load_plan(&plan, plan_path) == 0;
    if (had_plan) fprintf(stderr, "plan: %s\n", plan_path);

    csr_memcpy(&mat);
    real *d_x, *d_y;
    hipMalloc((void **)&d_x, sizeof(real) * (mat.N + MAX_BLOCK_SIZE));
    hipMalloc((void **)&d_y, sizeof(real) * (mat.M + WARP));
    hipMemset(d_x, 0, sizeof(real) * (mat.N + MAX_BLOCK_SIZE));
    hipMemcpy(d_x, x, sizeof(real) * mat.N, hipMemcpyHostToDevice);

    sfAMB amb;
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    sf_csr2amb(&amb, &mat, d_x, &plan);
    gettimeofday(&t1, NULL);
    if (keep_plan && !had_plan) (void)nsparse_save_plan(&plan, plan_path);
    printf("Format Conversion Cost (CSR=>AMB, %d-%d): %f[msec]\n", (int)amb.seg_size, amb.block_size,
           (float)(t1.tv_sec - t0.tv_sec) * 1000 + (float)(t1.tv_usec - t0.tv_usec) / 1000);

    hipEvent_t ev[2];
    hipEventCreate(&ev[0]);
    hipEventCreate(&ev[1]);
    float ave = 0;
    for (int i = 0; i < TRI_NUM; i++) {
        hipEventRecord(ev[0], 0);
        sf_spmv_amb(d_y, &amb, d_x, &plan);
        hipEventRecord(ev[1], 0);
        hipEventSynchronize(ev[1]);
        float ms = 0;
        hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (i > 0) ave += ms;
    }
    ave /= TRI_NUM - 1;
    hipMemcpy(y, d_y, sizeof(real) * mat.M, hipMemcpyDeviceToHost);
    printf("SpMV using AMB format: %s, %f[GFLOPS], %f[ms]\n", mat.matrix_name,
           (float)((double)mat.nnz * 2 / 1000 / 1000 / ave), ave);
    printf("AMB footprint: %lld bytes/SpMV => %f[GB/s]\n", nsparse_amb_footprint_bytes(&amb),
           (float)((double)nsparse_amb_footprint_bytes(&amb) / 1e6 / ave));

    hipFree(d_x);
    hipFree(d_y);
    release_amb(amb);
    release_csr(mat);
#ifdef sfDEBUG
    ans_check(csr_y, y, mat.M);
    free(csr_y);
#endif
    free(x);
    free(y);
    release_cpu_csr(mat);
    return 0;
}
