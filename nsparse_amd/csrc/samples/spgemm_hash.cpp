// spgemm_hash_{s,d} <file.mtx>  --  C = A * A with the hash SpGEMM.
// Same command line, timing protocol and output lines as the reference driver
// (cuda-c/src/sample/spgemm/spgemm_hash.cu:14-94): 11 runs, first discarded, GFLOPS from
// get_spgemm_flop.  The reference checks against cuSPARSE under sfDEBUG; cuSPARSE does not
// exist here, so the self-check is against a host Gustavson product written for this driver
// (NOT the test oracle; the product never links oracle/).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nsparse.h"

// Plain row-by-row product on the host, ascending columns: the "reference answer" role that
// spgemm_cu_csr (cuSPARSE) plays upstream (spgemm_hash.cu:60-68).
static void host_product(const sfCSR &a, const sfCSR &b, sfCSR *ans)
{
    const int M = a.M, N = b.N;
    std::vector<int> rpt(M + 1, 0), col;
    std::vector<real> val;
    std::vector<real> acc(N, 0);
    std::vector<int> stamp(N, -1), touched;
    for (int i = 0; i < M; i++) {
        touched.clear();
        for (int j = a.rpt[i]; j < a.rpt[i + 1]; j++) {
            const int k = a.col[j];
            for (int p = b.rpt[k]; p < b.rpt[k + 1]; p++) {
                const int c = b.col[p];
                if (stamp[c] != i) { stamp[c] = i; acc[c] = 0; touched.push_back(c); }
                acc[c] += a.val[j] * b.val[p];
            }
        }
        std::sort(touched.begin(), touched.end());
        for (int c : touched) { col.push_back(c); val.push_back(acc[c]); }
        rpt[i + 1] = (int)col.size();
    }
    ans->M = M; ans->N = N; ans->nnz = (int)col.size();
    ans->rpt = (int *)malloc(sizeof(int) * (M + 1));
    ans->col = (int *)malloc(sizeof(int) * (col.size() + 1));
    ans->val = (real *)malloc(sizeof(real) * (val.size() + 1));
    std::copy(rpt.begin(), rpt.end(), ans->rpt);
    std::copy(col.begin(), col.end(), ans->col);
    std::copy(val.begin(), val.end(), ans->val);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s matrix.mtx\n", argv[0]);
        return 1;
    }
    sfCSR a, b, c;
    init_csr_matrix_from_file(&a, argv[1]);
    init_csr_matrix_from_file(&b, argv[1]);
    csr_memcpy(&a);
    csr_memcpy(&b);

    long long flop = 0;
    get_spgemm_flop(&a, &b, a.M, &flop);

    hipEvent_t ev[2];
    hipEventCreate(&ev[0]);
    hipEventCreate(&ev[1]);
    float ave = 0;
    for (int i = 0; i < SPGEMM_TRI_NUM; i++) {
        if (i > 0) release_csr(c);
        hipEventRecord(ev[0], 0);
        spgemm_kernel_hash(&a, &b, &c);
        hipEventRecord(ev[1], 0);
        hipEventSynchronize(ev[1]);
        float ms = 0;
        hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (i > 0) ave += ms;
    }
    ave /= SPGEMM_TRI_NUM - 1;
    printf("SpGEMM using CSR format (Hash-based): %s, %f[GFLOPS], %f[ms]\n", a.matrix_name,
           (float)flop / 1000 / 1000 / ave, ave);

    csr_memcpyDtH(&c);
    release_csr(c);
#ifdef sfDEBUG
    sfCSR ans;
    host_product(a, b, &ans);
    printf("(nnz of A): %d =>\n(Num of intermediate products): %ld =>\n(nnz of C): %d\n", a.nnz,
           (long)(flop / 2), c.nnz);
    check_spgemm_answer(c, ans);
    release_cpu_csr(ans);
#endif
    release_csr(a);
    release_csr(b);
    release_cpu_csr(a);
    release_cpu_csr(b);
    release_cpu_csr(c);
    return 0;
}
