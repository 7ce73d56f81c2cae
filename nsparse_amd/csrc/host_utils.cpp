// host_utils.cpp -- host side of the drop-in boundary: MatrixMarket loader, plan record,
// CPU CSR SpMV, the two answer checks, host frees.  No device code in this file.
//
// Replaces (reference file:line):
//   init_csr_matrix_from_file / convert_file_csr   cuda-c/src/nsparse.cu:14-144
//   init_plan / set_plan                           cuda-c/src/nsparse.cu:171-187
//   init_vector                                    cuda-c/src/nsparse.cu:190-199
//   release_cpu_csr / release_cpu_amb              cuda-c/src/nsparse.cu:202-224
//   csr_kernel                                     cuda-c/src/nsparse.cu:240-259
//   ans_check / check_spgemm_answer                cuda-c/src/nsparse.cu:261-353
#include <sys/stat.h>

#include <cctype>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "internal.h"

namespace nsp {

// The error side channel is per calling thread: a caller reads the word of ITS last call, whatever other threads
// (one per GPU in amb_dist) did in between.  The product library starts no threads of its own.
static thread_local int g_err = 0;
static thread_local std::string g_err_msg;

void set_error(int code, const char *what, const char *file, int line)
{
    g_err = code;
    char buf[512];
    snprintf(buf, sizeof buf, "nsparse: error %d (%s) at %s:%d", code, what ? what : "?", file, line);
    g_err_msg = buf;
    fprintf(stderr, "%s\n", buf);
    const char *na = getenv("NSPARSE_NO_ABORT");
    if (!(na && na[0] == '1')) abort();
}
void clear_error()
{
    g_err = 0;
    g_err_msg.clear();
}

}  // namespace nsp

extern "C" {

int nsparse_save_csr_bin(const sfCSR *mat, const char *path);
int nsparse_load_csr_bin(sfCSR *mat, const char *path);

int nsparse_last_error(void) { return nsp::g_err; }
const char *nsparse_last_error_string(void) { return nsp::g_err_msg.c_str(); }

/* ---------------------------------------------------------------- loader --- */

// One COO triple as read from the file.
struct Entry {
    int r, c;
    real v;
};

// Parse a non-negative decimal; returns pointer past it, or nullptr when no digit.
static inline const char *parse_int(const char *p, const char *end, int *out)
{
    while (p < end && (*p == ' ' || *p == '\t')) p++;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = (*p == '-'); p++; }
    if (p >= end || !isdigit((unsigned char)*p)) return nullptr;
    long v = 0;
    while (p < end && isdigit((unsigned char)*p)) { v = v * 10 + (*p - '0'); p++; }
    *out = (int)(neg ? -v : v);
    return p;
}

static bool bin_cache_on()
{
    const char *e = getenv("NSPARSE_BIN_CACHE");
    return e && e[0] == '1';
}

void init_csr_matrix_from_file(sfCSR *mat, char *file_name)
{
    std::string bin_path = std::string(file_name) + ".csr.bin";
    if (bin_cache_on()) {
        struct stat sm, sb;
        if (stat(file_name, &sm) == 0 && stat(bin_path.c_str(), &sb) == 0 && sb.st_mtime >= sm.st_mtime &&
            nsparse_load_csr_bin(mat, bin_path.c_str()) == 0) {
            printf("Read mtx file: %s\n", file_name);  // same line as the text path
            mat->matrix_name = file_name;
            return;
        }
    }
    FILE *fp = fopen(file_name, "rb");
    if (fp == NULL) {
        printf("Cannot find file\n");  // reference nsparse.cu:35-38
        exit(1);
    }
    printf("Read mtx file: %s\n", file_name);  // reference nsparse.cu:39
    fseek(fp, 0, SEEK_END);
    long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<char> buf((size_t)fsz + 1);
    size_t got = fread(buf.data(), 1, (size_t)fsz, fp);
    fclose(fp);
    buf[got] = '\0';
    const char *p = buf.data(), *end = buf.data() + got;

    auto line_end = [&](const char *q) {
        const char *e = (const char *)memchr(q, '\n', (size_t)(end - q));
        return e ? e : end;
    };

    // banner: "general" anywhere in the first line => stored as is; anything else
    // (symmetric, skew-symmetric, hermitian, pattern symmetric) => mirrored with the
    // SAME sign (reference nsparse.cu:41-43,88-91,119-122).
    const char *le = line_end(p);
    bool unsym = std::string(p, le).find("general") != std::string::npos;
    p = le < end ? le + 1 : end;
    // skip comment lines, then the size line (reference nsparse.cu:44-49)
    while (p < end && *p == '%') { le = line_end(p); p = le < end ? le + 1 : end; }
    int M = 0, N = 0, nz_decl = 0;
    {
        const char *q = parse_int(p, end, &M);
        if (q) q = parse_int(q, end, &N);
        if (q) q = parse_int(q, end, &nz_decl);
        if (!q) {
            fprintf(stderr, "nsparse: malformed size line in %s\n", file_name);
            exit(1);
        }
        le = line_end(p);
        p = le < end ? le + 1 : end;
    }

    // entries: "row col [value]" per line; missing value => 1.0 (reference :68-76);
    // complex files: only the first value token is used.  The reference over-runs its
    // buffers if the file holds more than nz_decl entries; we stop at nz_decl.
    std::vector<Entry> coo;
    coo.reserve((size_t)(nz_decl > 0 ? nz_decl : 0));
    while (p < end && (int)coo.size() < nz_decl) {
        le = line_end(p);
        Entry e;
        const char *q = parse_int(p, le, &e.r);
        if (q) {
            q = parse_int(q, le, &e.c);
            if (q) {
                while (q < le && (*q == ' ' || *q == '\t')) q++;
                if (q < le && *q != '\r' && *q != '\n') {
                    e.v = (real)strtod(q, nullptr);  // atof semantics
                } else {
                    e.v = (real)1.0;
                }
                e.r -= 1;
                e.c -= 1;
                coo.push_back(e);
            }
        }
        p = le < end ? le + 1 : end;
    }
    buf.clear();
    buf.shrink_to_fit();

    // count, prefix, fill -- entry (r,c) is appended to row r and, when mirrored and
    // off-diagonal, (c,r) is appended to row c, in file order (reference :83-123).
    const int num = (int)coo.size();
    std::vector<int> cnt((size_t)(M > 0 ? M : 1), 0);
    long long total = num;
    for (int i = 0; i < num; i++) {
        cnt[coo[i].r]++;
        if (coo[i].c != coo[i].r && !unsym) { cnt[coo[i].c]++; total++; }
    }
    int *rpt = (int *)malloc(sizeof(int) * (size_t)(M + 1));
    int *col = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
    real *val = (real *)malloc(sizeof(real) * (size_t)(total > 0 ? total : 1));
    int off = 0, nnz_max = 0;
    for (int i = 0; i < M; i++) {
        rpt[i] = off;
        off += cnt[i];
        if (cnt[i] > nnz_max) nnz_max = cnt[i];
        cnt[i] = 0;  // reused as the per-row fill cursor
    }
    rpt[M] = off;
    for (int i = 0; i < num; i++) {
        const int r = coo[i].r, c = coo[i].c;
        col[rpt[r] + cnt[r]] = c;
        val[rpt[r] + cnt[r]++] = coo[i].v;
        if (c != r && !unsym) {
            col[rpt[c] + cnt[c]] = r;
            val[rpt[c] + cnt[c]++] = coo[i].v;
        }
    }
    mat->rpt = rpt;
    mat->col = col;
    mat->val = val;
    mat->M = M;
    mat->N = N;
    mat->nnz = (int)total;
    mat->nnz_max = nnz_max;
    mat->matrix_name = file_name;  // borrowed, like upstream (nsparse.cu:143)
    mat->d_rpt = nullptr;
    mat->d_col = nullptr;
    mat->d_val = nullptr;
    if (bin_cache_on()) (void)nsparse_save_csr_bin(mat, bin_path.c_str());
}

/* ------------------------------------------------- binary matrix cache ---- */
// parsing nlpkkt120's 1.5 GB of text dominates wall-clock.  A .csr.bin image
// holds exactly what the loader produces (same array contents, same in-row order), so loading
// it is interchangeable with parsing the .mtx.  With NSPARSE_BIN_CACHE=1 the loader itself
// reads `<file>.csr.bin` when it exists and is not older than the .mtx, and writes it otherwise.

namespace {
struct BinHeader {
    char magic[8];       // "NSPCSR01"
    int real_bytes;      // sizeof(real) of the writer
    int M, N, nnz, nnz_max;
    int reserved[3];
};
}  // namespace

int nsparse_save_csr_bin(const sfCSR *mat, const char *path)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    BinHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "NSPCSR01", 8);
    h.real_bytes = (int)sizeof(real);
    h.M = mat->M; h.N = mat->N; h.nnz = mat->nnz; h.nnz_max = mat->nnz_max;
    bool ok = fwrite(&h, sizeof h, 1, fp) == 1;
    ok = ok && fwrite(mat->rpt, sizeof(int), (size_t)mat->M + 1, fp) == (size_t)mat->M + 1;
    ok = ok && fwrite(mat->col, sizeof(int), (size_t)mat->nnz, fp) == (size_t)mat->nnz;
    ok = ok && fwrite(mat->val, sizeof(real), (size_t)mat->nnz, fp) == (size_t)mat->nnz;
    fclose(fp);
    return ok ? 0 : -2;
}

int nsparse_load_csr_bin(sfCSR *mat, const char *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    BinHeader h;
    if (fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, "NSPCSR01", 8) != 0 ||
        h.real_bytes != (int)sizeof(real) || h.M < 0 || h.nnz < 0) {
        fclose(fp);
        return -2;  // not ours / written by the other precision build
    }
    int *rpt = (int *)malloc(sizeof(int) * ((size_t)h.M + 1));
    int *col = (int *)malloc(sizeof(int) * (size_t)(h.nnz > 0 ? h.nnz : 1));
    real *val = (real *)malloc(sizeof(real) * (size_t)(h.nnz > 0 ? h.nnz : 1));
    bool ok = fread(rpt, sizeof(int), (size_t)h.M + 1, fp) == (size_t)h.M + 1;
    ok = ok && fread(col, sizeof(int), (size_t)h.nnz, fp) == (size_t)h.nnz;
    ok = ok && fread(val, sizeof(real), (size_t)h.nnz, fp) == (size_t)h.nnz;
    fclose(fp);
    if (!ok || rpt[h.M] != h.nnz) {
        free(rpt); free(col); free(val);
        return -3;
    }
    mat->rpt = rpt; mat->col = col; mat->val = val;
    mat->M = h.M; mat->N = h.N; mat->nnz = h.nnz; mat->nnz_max = h.nnz_max;
    mat->d_rpt = nullptr; mat->d_col = nullptr; mat->d_val = nullptr;
    return 0;
}

/* ------------------------------------------------------------------ plan --- */

int nsparse_save_plan(const sfPlan *plan, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    const int n = fprintf(fp, "NSPPLAN1 real %d chunk %d seg_size %zu block_size %d thread_block %zu thread_grid %zu "
                              "sigma %d seg_num %zu\n",
                          (int)sizeof(real), nsparse_set_amb_chunk(0), plan->seg_size, plan->block_size,
                          plan->thread_block, plan->thread_grid, plan->SIGMA, plan->seg_num);
    fclose(fp);
    return n > 0 ? 0 : -2;
}

int nsparse_load_plan(sfPlan *plan, const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int rb = 0, chunk = 0, bs = 0, sigma = 0;
    size_t seg = 0, tb = 0, tg = 0, sn = 0;
    const int got = fscanf(fp, "NSPPLAN1 real %d chunk %d seg_size %zu block_size %d thread_block %zu thread_grid %zu "
                               "sigma %d seg_num %zu",
                           &rb, &chunk, &seg, &bs, &tb, &tg, &sigma, &sn);
    fclose(fp);
    if (got != 8) return -2;
    if (rb != (int)sizeof(real) || chunk != nsparse_set_amb_chunk(0)) return -3;  // tuned for another build
    if (seg < 1 || seg > (size_t)USHORT_MAX || bs < 1 || bs > MAX_BLOCK_SIZE) return -4;
    plan->isPlan = TRUE;
    plan->seg_size = seg;
    plan->block_size = bs;
    plan->thread_block = tb;
    plan->thread_grid = tg;
    plan->SIGMA = sigma;
    plan->seg_num = sn;
    return 0;
}


void init_plan(sfPlan *plan) { plan->isPlan = FALSE; }

void set_plan(sfPlan *plan, size_t seg_size, int block_size)
{
    plan->isPlan = TRUE;
    plan->seg_size = seg_size > (size_t)USHORT_MAX ? (size_t)USHORT_MAX : seg_size;
    plan->block_size = (block_size < 1 || block_size > MAX_BLOCK_SIZE) ? 1 : block_size;
}

/* --------------------------------------------------------------- vectors --- */

void init_vector(real *x, int row)
{
    srand48((unsigned)time(NULL));
    for (int i = 0; i < row; i++) x[i] = (real)drand48();
}

void nsparse_init_vector_seeded(real *x, int row, unsigned long long seed)
{
    unsigned long long s = seed;
    for (int i = 0; i < row; i++) {
        unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (real)((double)(z >> 11) * (1.0 / 9007199254740992.0));
    }
}

/* ----------------------------------------------------------------- frees --- */

void release_cpu_csr(sfCSR mat)
{
    free(mat.rpt);
    free(mat.col);
    free(mat.val);
}

void release_cpu_amb(sfAMB mat)
{
    // The GPU path never fills the host mirrors (neither does upstream, where calling
    // this on a GPU-built sfAMB is undefined); free(NULL) is a no-op, so zero-initialised
    // structs are safe.
    free(mat.cs);
    free(mat.cl);
    free(mat.sellcs_val);
    free(mat.sellcs_col);
    free(mat.s_write_permutation);
    free(mat.s_write_permutation_offset);
}

/* ------------------------------------------------------------ CPU SpMV ---- */

void csr_kernel(real *y, sfCSR *cpu_mat, real *x)
{
    const int M = cpu_mat->M;
    const int *rpt = cpu_mat->rpt, *col = cpu_mat->col;
    const real *val = cpu_mat->val;
    for (int i = 0; i < M; i++) {
        real acc = 0;
        const int b = rpt[i], e = rpt[i + 1];
        for (int j = b; j < e; j++) acc += val[j] * x[col[j]];
        y[i] = acc;
    }
}

/* ---------------------------------------------------------------- checks --- */

static inline real rabs(real v) { return v < 0 ? -v : v; }

#if NSPARSE_REAL_IS_FLOAT
static const real kScale = 1000;
#else
static const real kScale = 1000 * 1000;
#endif

int nsparse_ans_check_count(const real *csr_ans, const real *ans_vec, int N)
{
    int fails = 0;
    for (int i = 0; i < N; i++)
        if (rabs(ans_vec[i] - csr_ans[i]) * 100 * kScale > rabs(ans_vec[i])) fails++;
    return fails;
}

void ans_check(real *csr_ans, real *ans_vec, int N)
{
    int shown = 0;
    for (int i = 0; i < N && shown < 10; i++) {
        const real delta = rabs(ans_vec[i] - csr_ans[i]);
        if (delta * 100 * kScale > rabs(ans_vec[i])) {
            printf("i=%d, ans=%e, csr=%e, delta=%e\n", i, (double)ans_vec[i], (double)csr_ans[i],
                   (double)delta);
            shown++;
        }
    }
    printf(shown ? "Calculation Result is Incorrect\n" : "Calculation Result is Correct\n");
}

int nsparse_check_spgemm_count(const sfCSR *c, const sfCSR *ans)
{
    if (c->nnz != ans->nnz) return -1;
    for (int i = 0; i <= c->M; i++)
        if (c->rpt[i] != ans->rpt[i]) return -2;
    for (int i = 0; i < c->nnz; i++)
        if (c->col[i] != ans->col[i]) return -3;
    int fails = 0;
    for (int i = 0; i < c->nnz; i++)
        if (rabs(ans->val[i] - c->val[i]) * 1000 * kScale > rabs(ans->val[i])) fails++;
    return fails;
}

void check_spgemm_answer(sfCSR c, sfCSR ans)
{
    if (c.nnz != ans.nnz) {
        printf("nnz is not correct: %d (correct), %d (incorrect)\n", ans.nnz, c.nnz);
        return;
    }
    for (int i = 0; i <= c.M; i++) {
        if (c.rpt[i] != ans.rpt[i]) {
            printf("rpt[%d] is not correct: %d (correct),%d (incorrect)\n", i, ans.rpt[i], c.rpt[i]);
            return;
        }
    }
    for (int i = 0; i < c.nnz; i++) {
        if (c.col[i] != ans.col[i]) {
            printf("col[%d] is not correct: %d (correct), %d (incorrect)\n", i, ans.col[i], c.col[i]);
            return;
        }
    }
    int shown = 0;
    for (int i = 0; i < c.nnz && shown < 10; i++) {
        const real delta = rabs(ans.val[i] - c.val[i]);
        if (delta * 1000 * kScale > rabs(ans.val[i])) {
            printf("val[%d]: ans=%e, c=%e, delta=%e\n", i, (double)ans.val[i], (double)c.val[i],
                   (double)delta);
            shown++;
        }
    }
    printf(shown ? "Calculation Result is Incorrect\n" : "Calculation Result is Correct\n");
}


#ifndef NSPARSE_SRC_HASH
#define NSPARSE_SRC_HASH "unknown"
#endif
// "gfx950 <precision> <content hash of csrc + headers at build time>[ experiments]": lets a harness check that the
// shared object it loaded was built from the sources beside it, and whether it is the -DNSPARSE_EXPERIMENTS variant
// (the one that reads the measurement switches and carries the opt-in kernels).
#ifdef NSPARSE_EXPERIMENTS
#define NSP_BUILD_KIND " experiments"
#else
#define NSP_BUILD_KIND ""
#endif
const char *nsparse_build_info(void)
{
    return NSPARSE_REAL_IS_FLOAT ? "gfx950 float " NSPARSE_SRC_HASH NSP_BUILD_KIND : "gfx950 double " NSPARSE_SRC_HASH NSP_BUILD_KIND;
}

}  // extern "C"
