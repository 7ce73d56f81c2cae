// synth.cpp -- deterministic synthetic stand-ins for the SuiteSparse inputs of BASELINE.md
// (no network on the GPU box, so cant / webbase-1M / nlpkkt120 cannot be downloaded) and the
// R-MAT generator of config 5.  Host only.  Columns ascend inside every row, as they do for a
// column-major-sorted .mtx through the reference loader.
//
// There is no reference counterpart: the reference reads .mtx files only.  These generators
// exist so that bench.py and the full-size property tests have the same inputs on every box.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "internal.h"

namespace {

inline unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double u01(unsigned long long z) { return (double)(z >> 11) * (1.0 / 9007199254740992.0); }

// symmetric pseudo-random value for the unordered pair (r, c)
inline real pair_value(unsigned long long seed, long long r, long long c)
{
    const long long lo = r < c ? r : c, hi = r < c ? c : r;
    const double u = u01(mix64(seed ^ mix64((unsigned long long)lo * 0x100000001B3ull + (unsigned long long)hi)));
    return (real)(r == c ? 4.0 + u : 0.1 + u);
}

struct Builder {
    std::vector<int> rpt, col;
    std::vector<real> val;
    int nnz_max = 0;
    void finish(sfCSR *mat, long long M, long long N)
    {
        mat->M = (int)M;
        mat->N = (int)N;
        mat->nnz = (int)col.size();
        mat->nnz_max = nnz_max;
        mat->rpt = (int *)malloc(sizeof(int) * (size_t)(M + 1));
        mat->col = (int *)malloc(sizeof(int) * (col.size() ? col.size() : 1));
        mat->val = (real *)malloc(sizeof(real) * (val.size() ? val.size() : 1));
        memcpy(mat->rpt, rpt.data(), sizeof(int) * (size_t)(M + 1));
        if (!col.empty()) {
            memcpy(mat->col, col.data(), sizeof(int) * col.size());
            memcpy(mat->val, val.data(), sizeof(real) * val.size());
        }
        mat->d_rpt = nullptr;
        mat->d_col = nullptr;
        mat->d_val = nullptr;
        mat->matrix_name = (char *)"synthetic";
    }
};

// 27-point stencil on an nx*ny*nz grid with `dof` unknowns per node (dof=3: FEM brick,
// "cant" class; dof=1: scalar grid, "nlpkkt" class).  Global ids: node*dof + d.
void gen_stencil(sfCSR *mat, int dof, long long nx, long long ny, long long nz,
                 unsigned long long seed, long long rb, long long re)
{
    const long long nodes = nx * ny * nz, Mfull = nodes * dof;
    if (re <= 0 || re > Mfull) re = Mfull;
    if (rb < 0) rb = 0;
    Builder b;
    const long long M = re - rb;
    b.rpt.resize((size_t)M + 1);
    b.col.reserve((size_t)M * 27 * dof);
    b.val.reserve((size_t)M * 27 * dof);
    for (long long r = rb; r < re; r++) {
        b.rpt[(size_t)(r - rb)] = (int)b.col.size();
        const long long node = r / dof;
        const long long x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        const size_t before = b.col.size();
        for (long long dz = -1; dz <= 1; dz++) {
            const long long zz = z + dz;
            if (zz < 0 || zz >= nz) continue;
            for (long long dy = -1; dy <= 1; dy++) {
                const long long yy = y + dy;
                if (yy < 0 || yy >= ny) continue;
                for (long long dx = -1; dx <= 1; dx++) {
                    const long long xx = x + dx;
                    if (xx < 0 || xx >= nx) continue;
                    const long long nb = (zz * ny + yy) * nx + xx;
                    for (int d = 0; d < dof; d++) {
                        const long long c = nb * dof + d;
                        b.col.push_back((int)c);
                        b.val.push_back(pair_value(seed, r, c));
                    }
                }
            }
        }
        const int len = (int)(b.col.size() - before);
        if (len > b.nnz_max) b.nnz_max = len;
    }
    b.rpt[(size_t)M] = (int)b.col.size();
    b.finish(mat, M, Mfull);
}

// power-law web graph ("webbase" class): short rows with a heavy tail of long ones, columns
// split between a local window and globally popular hub pages.
void gen_powerlaw(sfCSR *mat, long long n, long long target_nnz, unsigned long long seed,
                  long long rb, long long re)
{
    if (re <= 0 || re > n) re = n;
    if (rb < 0) rb = 0;
    const double avg = (double)target_nnz / (double)n;
    Builder b;
    const long long M = re - rb;
    b.rpt.resize((size_t)M + 1);
    std::vector<int> tmp;
    for (long long r = rb; r < re; r++) {
        b.rpt[(size_t)(r - rb)] = (int)b.col.size();
        unsigned long long s = mix64(seed ^ mix64((unsigned long long)r));
        // Pareto(alpha = 2.1) degree with mean ~avg, capped at n/200
        const double u = u01(s);
        double d = (avg * 0.52) / std::pow(1.0 - u, 1.0 / 2.1);
        long long deg = (long long)d;
        if (deg < 1) deg = 1;
        if (deg > n / 200) deg = n / 200;
        tmp.clear();
        for (long long k = 0; k < deg; k++) {
            s = mix64(s);
            const double a = u01(s);
            s = mix64(s);
            long long c;
            if (a < 0.55) {  // local link
                c = r + (long long)((u01(s) - 0.5) * 2000.0);
            } else {  // hub: heavily skewed towards low ids
                const double t = u01(s);
                c = (long long)((double)n * t * t * t * t);
            }
            if (c < 0) c = 0;
            if (c >= n) c = n - 1;
            tmp.push_back((int)c);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (int c : tmp) {
            b.col.push_back(c);
            b.val.push_back((real)(0.1 + u01(mix64(seed ^ mix64((unsigned long long)r * 0x9E3779B1ull + (unsigned long long)c)))));
        }
        if ((int)tmp.size() > b.nnz_max) b.nnz_max = (int)tmp.size();
    }
    b.rpt[(size_t)M] = (int)b.col.size();
    b.finish(mat, M, n);
}

// "cant class, irregular" (kind 5): the same nx*ny*nz brick of 3-dof nodes as kind 0, but
//   * the unknowns are renumbered by a pseudo-random permutation inside consecutive blocks of
//     `blk` unknowns = three mesh planes (symmetric: P A P^T), so the rows of one node are no longer neighbours, the
//     column window of a C row is no longer the 5 planes a natural ordering gives, and rows with
//     the same column pattern are not adjacent;
//   * node couplings are dropped (symmetrically, all 3 x 3 dof at once) with probability `drop`,
//     which brings nnz and the product count down to the SuiteSparse statistics of cant
//     (62,451 rows, 4.0 M nnz, ~0.27 G products for 9 x 9 x 257 with drop = 0.074).
// Values as in kind 0 (symmetric in the ORIGINAL numbering, so the matrix stays symmetric).
//
// kind 6 = kind 5 + SCALAR perturbations (`permille` of the nodes, chosen by hash): what boundary conditions
// and mixed elements do to a real finite-element matrix, and what the all-3x3-at-once drops above leave
// intact -- the rows of one node keep one column pattern.  A chosen node either gets one of its dof
// CONSTRAINED (Dirichlet: the row keeps its diagonal only, the column disappears from every other row) or
// loses ONE scalar coupling (d_a of the node, d_b of one neighbour node; symmetric), so that row no longer
// has the pattern of its node mates.
struct Perturb {
    unsigned long long seed;
    long long nx, ny, nz;
    unsigned long long thr;  // 0: off
    // what happens at `node`: kind 0 nothing, 1 constrained dof `da`, 2 coupling (da, nb:db) dropped
    struct What { int kind, da, db; long long nb; };
    What at(long long node) const
    {
        What w = {0, 0, 0, -1};
        if (!thr) return w;
        const unsigned long long h = mix64(seed ^ mix64(0xC0A5ull + (unsigned long long)node));
        if (h >= thr) return w;
        const unsigned long long g = mix64(h);
        w.da = (int)((g >> 16) % 3);
        if ((g >> 8) & 1) {
            w.kind = 1;
            return w;
        }
        const int k = (int)((g >> 20) % 26);
        const int q = k < 13 ? k : k + 1;  // 0..26 without the centre (13)
        const long long x = node % nx + (q % 3 - 1), y = (node / nx) % ny + ((q / 3) % 3 - 1), z = node / (nx * ny) + (q / 9 - 1);
        if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) return w;
        w.kind = 2;
        w.db = (int)((g >> 28) % 3);
        w.nb = (z * ny + y) * nx + x;
        return w;
    }
};

void gen_brick_shuffled(sfCSR *mat, long long nx, long long ny, long long nz, unsigned long long seed,
                        long long rb, long long re, int permille = 0)
{
    const int dof = 3;
    const Perturb pert = {seed, nx, ny, nz,
                          permille >= 1000 ? ~0ull : (unsigned long long)((double)permille / 1000.0 * 18446744073709551615.0)};
    const long long nodes = nx * ny * nz, Mfull = nodes * dof;
    if (re <= 0 || re > Mfull) re = Mfull;
    if (rb < 0) rb = 0;
    const long long blk = 3 * dof * nx * ny;  // three planes of unknowns
    const unsigned long long drop_thr = (unsigned long long)(0.074 * 18446744073709551615.0);
    // new_of[old] / old_of[new]: permutation inside each block, by sorting hash keys
    std::vector<int> new_of((size_t)Mfull), old_of((size_t)Mfull);
    {
        std::vector<std::pair<unsigned long long, int>> keys;
        for (long long b0 = 0; b0 < Mfull; b0 += blk) {
            const long long b1 = b0 + blk < Mfull ? b0 + blk : Mfull;
            keys.clear();
            for (long long i = b0; i < b1; i++) keys.emplace_back(mix64(seed ^ mix64(0xB10Cull + (unsigned long long)i)), (int)i);
            std::sort(keys.begin(), keys.end());
            for (long long i = b0; i < b1; i++) {
                old_of[(size_t)i] = keys[(size_t)(i - b0)].second;
                new_of[(size_t)keys[(size_t)(i - b0)].second] = (int)i;
            }
        }
    }
    Builder b;
    const long long M = re - rb;
    b.rpt.resize((size_t)M + 1);
    b.col.reserve((size_t)M * 27 * dof);
    b.val.reserve((size_t)M * 27 * dof);
    std::vector<std::pair<int, real>> row;
    for (long long rn = rb; rn < re; rn++) {
        b.rpt[(size_t)(rn - rb)] = (int)b.col.size();
        const long long r = old_of[(size_t)rn];
        const long long node = r / dof;
        const long long x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        row.clear();
        const int d_row = (int)(r % dof);
        const Perturb::What mine = pert.at(node);
        if (mine.kind == 1 && mine.da == d_row) {  // constrained unknown: diagonal only
            b.col.push_back((int)rn);
            b.val.push_back(pair_value(seed, r, r));
            if (b.nnz_max < 1) b.nnz_max = 1;
            continue;
        }
        for (long long dz = -1; dz <= 1; dz++) {
            const long long zz = z + dz;
            if (zz < 0 || zz >= nz) continue;
            for (long long dy = -1; dy <= 1; dy++) {
                const long long yy = y + dy;
                if (yy < 0 || yy >= ny) continue;
                for (long long dx = -1; dx <= 1; dx++) {
                    const long long xx = x + dx;
                    if (xx < 0 || xx >= nx) continue;
                    const long long nb = (zz * ny + yy) * nx + xx;
                    if (nb != node) {
                        const long long lo = nb < node ? nb : node, hi = nb < node ? node : nb;
                        if (mix64(seed ^ mix64(0xD40Full + (unsigned long long)lo * 0x100000001B3ull + (unsigned long long)hi)) < drop_thr)
                            continue;
                    }
                    const Perturb::What theirs = nb == node ? mine : pert.at(nb);
                    for (int d = 0; d < dof; d++) {
                        const long long c = nb * dof + d;
                        if (theirs.kind == 1 && theirs.da == d) continue;                           // column of a constrained unknown
                        if (mine.kind == 2 && mine.nb == nb && mine.da == d_row && mine.db == d) continue;      // dropped scalar coupling
                        if (theirs.kind == 2 && theirs.nb == node && theirs.da == d && theirs.db == d_row) continue;  // ... seen from the other side
                        row.emplace_back(new_of[(size_t)c], pair_value(seed, r, c));
                    }
                }
            }
        }
        std::sort(row.begin(), row.end());
        for (auto &e : row) {
            b.col.push_back(e.first);
            b.val.push_back(e.second);
        }
        if ((int)row.size() > b.nnz_max) b.nnz_max = (int)row.size();
    }
    b.rpt[(size_t)M] = (int)b.col.size();
    b.finish(mat, M, Mfull);
}

// Web graph with the SuiteSparse statistics of webbase-1M (kind 4; BASELINE config 3):
// 1,000,005 pages, 3.1 M links, ~70 M intermediate products and ~51 M non-zeros in A^2, longest
// row ~4.7 K.  Kind 2 has the row-length distribution only: its products are 11x fewer than the
// real matrix's because its long rows are not the popular ones.  Here pages come in sites of
// kSite consecutive ids; the first page of a site is its index page with a heavy-tailed number
// of links (directories), index pages of popular sites are long AND linked from everywhere, so
// the popular columns select long rows of B -- which is what makes n_prod / nnz = 22 in
// webbase-1M.  Every row depends on its own id and the seed only (row blocks can be generated
// independently).
constexpr long long kSite = 32;
inline long long web_index_degree(unsigned long long seed, long long site, long long n_sites, long long cap)
{
    // popularity rank of a site = its id (low ids popular); degree: Pareto tail, longer for popular sites
    const double u = u01(mix64(seed ^ mix64(0x51DEull + (unsigned long long)site)));
    const double pop = 1.0 - (double)site / (double)n_sites;  // 1 = most popular
    double d = (5.0 + 36.0 * pop * pop * pop) / std::pow(1.0 - u, 1.0 / 1.5);
    if (d > (double)cap) d = (double)cap;
    return (long long)d;
}
void gen_webgraph(sfCSR *mat, long long n, long long target_nnz, unsigned long long seed, long long rb,
                  long long re)
{
    if (re <= 0 || re > n) re = n;
    if (rb < 0) rb = 0;
    const long long n_sites = (n + kSite - 1) / kSite;
    const long long cap = 4700;
    // ordinary pages: 1 + geometric, mean = the share `ofr` of target_nnz (index pages and template
    // sites carry the rest)
    // constants tuned (tools: the statistics line of tests/test_configs_gpu.py) until nnz, the product
    // count and nnz(A^2) land within 2 % of webbase-1M's 3,105,536 / 69.5 M / 51.1 M
    constexpr double ofr = 0.47, skp = 3.4, i_loc = 0.70, i_dir = 0.90, o_home = 0.30, o_loc = 0.55, o_dir = 0.75,
                     wmul = 2.0, tpl = 0.03;
    const double ord_mean = ((double)target_nnz * ofr) / (double)(n - n_sites);
    const double pgeo = ord_mean > 1.0 ? 1.0 - 1.0 / ord_mean : 0.0;
    Builder b;
    const long long M = re - rb;
    b.rpt.resize((size_t)M + 1);
    std::vector<int> tmp;
    auto skew_site = [&](double t) {  // popular sites (low ids) drawn far more often
        long long s = (long long)((double)n_sites * std::pow(t, skp));
        return s >= n_sites ? n_sites - 1 : s;
    };
    for (long long r = rb; r < re; r++) {
        b.rpt[(size_t)(r - rb)] = (int)b.col.size();
        unsigned long long s = mix64(seed ^ mix64((unsigned long long)r));
        const long long site = r / kSite;
        const bool is_index = r % kSite == 0;
        long long deg;
        const unsigned long long sh = mix64(seed ^ mix64(0x7E3Full + (unsigned long long)site));
        if (u01(sh) < tpl && (site + 1) * kSite <= n) {
            // template site: every page carries the same navigation bar of K same-site links
            const long long K = 16 + (long long)((sh >> 7) % 17);
            for (long long k = 0; k < K; k++) {
                const long long c = site * kSite + k;
                b.col.push_back((int)c);
                b.val.push_back((real)(0.1 + u01(mix64(seed ^ mix64((unsigned long long)r * 0x9E3779B1ull + (unsigned long long)c)))));
            }
            if ((int)K > b.nnz_max) b.nnz_max = (int)K;
            continue;
        }
        if (is_index) {
            deg = web_index_degree(seed, site, n_sites, cap);
        } else {
            deg = 1;
            while (deg < 40 && u01(s = mix64(s)) < pgeo) deg++;
        }
        tmp.clear();
        for (long long k = 0; k < deg; k++) {
            s = mix64(s);
            const double a = u01(s);
            s = mix64(s);
            const double t = u01(s);
            long long c;
            if (is_index) {
                if (a < i_loc) {  // the pages of the site and of the sites next to it
                    const long long w = deg < kSite ? kSite : (long long)(wmul * deg);
                    c = site * kSite + (long long)(t * (double)w) - (w - kSite) / 2;
                } else if (a < i_dir) {  // other directories
                    c = skew_site(t) * kSite;
                } else {
                    c = (long long)(t * (double)n);
                }
            } else {
                if (a < o_home) c = site * kSite;                               // home
                else if (a < o_loc) c = site * kSite + (long long)(t * kSite); // a page of the site
                else if (a < o_dir) c = skew_site(t) * kSite;                  // a popular directory
                else c = (long long)(t * (double)n);
            }
            if (c < 0) c = 0;
            if (c >= n) c = n - 1;
            tmp.push_back((int)c);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (int c : tmp) {
            b.col.push_back(c);
            b.val.push_back((real)(0.1 + u01(mix64(seed ^ mix64((unsigned long long)r * 0x9E3779B1ull + (unsigned long long)c)))));
        }
        if ((int)tmp.size() > b.nnz_max) b.nnz_max = (int)tmp.size();
    }
    b.rpt[(size_t)M] = (int)b.col.size();
    b.finish(mat, M, n);
}

// R-MAT (a,b,c,d) = (0.57,0.19,0.19,0.05), no vertex permutation, duplicates merged with
// summed values.
void gen_rmat(sfCSR *mat, int scale, long long ef, long long edges, unsigned long long seed, long long rb,
              long long re)
{
    const long long n = 1LL << scale, m = edges > 0 ? edges : n * ef;
    if (re <= 0 || re > n) re = n;
    if (rb < 0) rb = 0;
    std::vector<unsigned long long> e;
    e.reserve((size_t)m);
    unsigned long long s = mix64(seed);
    for (long long k = 0; k < m; k++) {
        unsigned long long r = 0, c = 0;
        for (int bit = 0; bit < scale; bit++) {
            s = mix64(s);
            const double u = u01(s);
            const int rbit = u >= 0.76;                               // c or d quadrant
            const int cbit = (u >= 0.57 && u < 0.76) || (u >= 0.95);  // b or d quadrant
            r |= (unsigned long long)rbit << bit;
            c |= (unsigned long long)cbit << bit;
        }
        if ((long long)r >= rb && (long long)r < re) e.push_back((r << 32) | c);
    }
    std::sort(e.begin(), e.end());
    Builder b;
    const long long M = re - rb;
    b.rpt.assign((size_t)M + 1, 0);
    size_t i = 0;
    for (long long r = rb; r < re; r++) {
        b.rpt[(size_t)(r - rb)] = (int)b.col.size();
        const size_t before = b.col.size();
        while (i < e.size() && (long long)(e[i] >> 32) == r) {
            const unsigned long long key = e[i];
            int mult = 0;
            while (i < e.size() && e[i] == key) { mult++; i++; }
            const int c = (int)(key & 0xffffffffull);
            b.col.push_back(c);
            b.val.push_back((real)(mult * (0.05 + u01(mix64(seed ^ mix64(key))))));
        }
        const int len = (int)(b.col.size() - before);
        if (len > b.nnz_max) b.nnz_max = len;
    }
    b.rpt[(size_t)M] = (int)b.col.size();
    b.finish(mat, M, n);
}

}  // namespace

extern "C" void nsparse_synth_csr(sfCSR *mat, int kind, long long p0, long long p1, long long p2,
                                  unsigned long long seed, long long row_begin, long long row_end)
{
    switch (kind) {
        case 0: gen_stencil(mat, 3, p0, p1, p2, seed, row_begin, row_end); break;
        case 1: gen_stencil(mat, 1, p0, p1, p2, seed, row_begin, row_end); break;
        case 2: gen_powerlaw(mat, p0, p1, seed, row_begin, row_end); break;
        case 3: gen_rmat(mat, (int)p0, p1, p2, seed, row_begin, row_end); break;
        case 4: gen_webgraph(mat, p0, p1, seed, row_begin, row_end); break;
        case 5: gen_brick_shuffled(mat, p0, p1, p2, seed, row_begin, row_end); break;
        // kind 6: p2 = nz + (permille of the nodes that get a scalar perturbation) * 2^32
        case 6: gen_brick_shuffled(mat, p0, p1, p2 & 0xffffffffLL, seed, row_begin, row_end, (int)(p2 >> 32)); break;
        default:
            fprintf(stderr, "nsparse_synth_csr: unknown kind %d\n", kind);
            memset(mat, 0, sizeof(*mat));
    }
}

// A host CSR written as a Matrix Market file the way the SuiteSparse collection ships its matrices
// (coordinate format, 1-based, entries sorted by column then row, `symmetric` files holding the lower
// triangle only), so that the loader (init_csr_matrix_from_file, reference nsparse.cu:14-136) and the sample
// drivers can be exercised at full size on a box without network.  flavour 0: real general, 1: real
// symmetric (the matrix must be symmetric: only entries with row >= column are written), 2: pattern
// general, 3: pattern symmetric.  Values are printed with enough digits to read back bit-identical.
extern "C" int nsparse_write_mtx(const sfCSR *m, const char *path, int flavour)
{
    if (!m || !path || flavour < 0 || flavour > 3) return -1;
    FILE *f = fopen(path, "w");
    if (!f) return -2;
    const bool sym = flavour == 1 || flavour == 3, pat = flavour >= 2;
    std::vector<char> buf(1 << 22);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    const char *vfmt = sizeof(real) == 8 ? "%d %d %.17g\n" : "%d %d %.9g\n";
    long long stored = 0;
    if (sym) {
        for (int i = 0; i < m->M; i++)
            for (int j = m->rpt[i]; j < m->rpt[i + 1]; j++) stored += m->col[j] >= i;
    } else {
        stored = m->nnz;
    }
    fprintf(f, "%%%%MatrixMarket matrix coordinate %s %s\n", pat ? "pattern" : "real", sym ? "symmetric" : "general");
    fprintf(f, "%%-------------------------------------------------------------------------------\n");
    fprintf(f, "%% nsparse_write_mtx: stand-in written in SuiteSparse conventions (column-major sorted%s)\n",
            sym ? ", lower triangle" : "");
    fprintf(f, "%%-------------------------------------------------------------------------------\n");
    fprintf(f, "%d %d %lld\n", m->M, m->N, stored);
    if (sym) {
        // column c of the lower triangle = the entries (r >= c, c) = by symmetry row c of the CSR from the
        // diagonal on: rows of the CSR are walked in order, every entry (i, j >= i) leaves as "j i"
        for (int i = 0; i < m->M; i++)
            for (int j = m->rpt[i]; j < m->rpt[i + 1]; j++) {
                if (m->col[j] < i) continue;
                if (pat) fprintf(f, "%d %d\n", m->col[j] + 1, i + 1);
                else fprintf(f, vfmt, m->col[j] + 1, i + 1, (double)m->val[j]);
            }
    } else {
        // counting sort by column (stable: rows ascend inside a column)
        std::vector<long long> start((size_t)m->N + 1, 0);
        for (int j = 0; j < m->nnz; j++) start[(size_t)m->col[j] + 1]++;
        for (int c = 0; c < m->N; c++) start[(size_t)c + 1] += start[(size_t)c];
        std::vector<int> row_of((size_t)m->nnz), src((size_t)m->nnz);
        for (int i = 0; i < m->M; i++)
            for (int j = m->rpt[i]; j < m->rpt[i + 1]; j++) {
                const long long p = start[(size_t)m->col[j]]++;
                row_of[(size_t)p] = i;
                src[(size_t)p] = j;
            }
        for (long long p = 0; p < m->nnz; p++) {
            const int j = src[(size_t)p];
            if (pat) fprintf(f, "%d %d\n", row_of[(size_t)p] + 1, m->col[j] + 1);
            else fprintf(f, vfmt, row_of[(size_t)p] + 1, m->col[j] + 1, (double)m->val[j]);
        }
    }
    const int rc = ferror(f) ? -3 : 0;
    fclose(f);
    return rc;
}
