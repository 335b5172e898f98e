// mtx.cpp -- Matrix Market files <-> the canonical host matrix (host-only part of libvbnmf_hip.so).
//
// Replaces, for the count matrix, `as(Matrix::readMM(count), 'dgCMatrix')` of read_10x (reference
// R/utils.R:34) and `Matrix::writeMM` of write_10x (reference R/utils.R:876): the file goes straight into
// the compressed-column form the engine ingests, never through a dense or triplet R object.  The format is
// the published NIST one (https://math.nist.gov/MatrixMarket/formats.html) that Matrix::readMM implements:
//   %%MatrixMarket matrix coordinate {real|integer|pattern} {general|symmetric|skew-symmetric}
//   % comments ...
//   rows cols entries
//   i j [value]            1-based; repeated (i, j) are summed, as dgTMatrix -> dgCMatrix does
// and the dense `array` form (column-major values, general or symmetric lower triangle).
// The body is parsed by all host threads: the mapped file is cut at line starts, each thread converts its
// piece to triplets, and a counting sort by column feeds the same canonicaliser the CSC entry point uses.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstring>

#include "common.h"

namespace vbnmf {
namespace {

struct Mapped {
    const char *p = nullptr;
    size_t len = 0;
    int fd = -1;
    ~Mapped()
    {
        if (p && len) munmap(const_cast<char *>(p), len);
        if (fd >= 0) close(fd);
    }
};

inline const char *skip_space(const char *s, const char *end)
{
    while (s < end && (*s == ' ' || *s == '\t' || *s == '\r')) s++;
    return s;
}

// Unsigned decimal integer; returns nullptr when no digit is there.
inline const char *parse_uint(const char *s, const char *end, int64_t &v)
{
    s = skip_space(s, end);
    if (s < end && *s == '+') s++;
    if (s >= end || *s < '0' || *s > '9') return nullptr;
    int64_t x = 0;
    while (s < end && *s >= '0' && *s <= '9') { x = x * 10 + (*s - '0'); s++; }
    v = x;
    return s;
}

// A real number: plain digit strings (the usual count file) take the fast path, everything else strtod.
inline const char *parse_real(const char *s, const char *end, double &v)
{
    s = skip_space(s, end);
    const char *q = s;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    const char *d0 = q;
    int64_t x = 0;
    while (q < end && *q >= '0' && *q <= '9' && q - d0 < 18) { x = x * 10 + (*q - '0'); q++; }
    if (q > d0 && (q == end || *q == ' ' || *q == '\t' || *q == '\r' || *q == '\n')) {
        v = neg ? -(double)x : (double)x;
        return q;
    }
    char buf[64];
    size_t n = 0;
    while (s + n < end && n < sizeof(buf) - 1 && s[n] != ' ' && s[n] != '\t' && s[n] != '\r' && s[n] != '\n') { buf[n] = s[n]; n++; }
    if (n == 0) return nullptr;
    buf[n] = 0;
    char *e = nullptr;
    v = std::strtod(buf, &e);
    if (e == buf) return nullptr;
    return s + (e - buf);
}

inline const char *next_line(const char *s, const char *end)
{
    const char *nl = (const char *)memchr(s, '\n', end - s);
    return nl ? nl + 1 : end;
}

std::string lower_token(const char *&s, const char *end)
{
    s = skip_space(s, end);
    std::string t;
    while (s < end && !isspace((unsigned char)*s)) { t.push_back((char)tolower((unsigned char)*s)); s++; }
    return t;
}

struct Piece {
    std::vector<int32_t> row, col;
    std::vector<double> val;
    int64_t lines = 0;
    int bad = 0;                   // 1 malformed line, 2 index out of range
    int64_t bad_at = 0;            // byte offset of the offending line
};

}  // namespace

int matrix_from_mtx(const char *path, Matrix &X)
{
    Mapped f;
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) return fail(VBNMF_ERR_BAD_ARG, "Count file %s does not exist (%s)", path, strerror(errno));   // R/utils.R:33
    struct stat st;
    if (fstat(f.fd, &st) != 0 || st.st_size <= 0) return fail(VBNMF_ERR_BAD_ARG, "%s is empty or unreadable", path);
    f.len = (size_t)st.st_size;
    void *mp = mmap(nullptr, f.len, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (mp == MAP_FAILED) { f.len = 0; return fail(VBNMF_ERR_OOM, "mmap of %s failed: %s", path, strerror(errno)); }
    f.p = (const char *)mp;
    const char *s = f.p, *end = f.p + f.len;

    // banner
    const char *ls = s, *le = next_line(s, end);
    std::string banner = lower_token(ls, le);
    if (banner != "%%matrixmarket") return fail(VBNMF_ERR_BAD_ARG, "%s: not a Matrix Market file (no %%%%MatrixMarket banner)", path);
    const std::string object = lower_token(ls, le), format = lower_token(ls, le), field = lower_token(ls, le),
                      symmetry = lower_token(ls, le);
    if (object != "matrix") return fail(VBNMF_ERR_BAD_ARG, "%s: object '%s' is not 'matrix'", path, object.c_str());
    const bool coordinate = format == "coordinate";
    if (!coordinate && format != "array") return fail(VBNMF_ERR_BAD_ARG, "%s: format '%s' is neither coordinate nor array", path, format.c_str());
    const bool pattern = field == "pattern";
    if (!pattern && field != "real" && field != "integer" && field != "double")
        return fail(VBNMF_ERR_BAD_ARG, "%s: field '%s' is not real, integer or pattern", path, field.c_str());
    if (pattern && !coordinate) return fail(VBNMF_ERR_BAD_ARG, "%s: a pattern matrix must be in coordinate format", path);
    int sym = 0;                   // 0 general, 1 symmetric, 2 skew-symmetric
    if (symmetry == "symmetric") sym = 1;
    else if (symmetry == "skew-symmetric") sym = 2;
    else if (symmetry != "general") return fail(VBNMF_ERR_BAD_ARG, "%s: symmetry '%s' is not supported", path, symmetry.c_str());
    s = le;
    // comments and blank lines, then the size line
    while (s < end) {
        const char *t = skip_space(s, end);
        if (t < end && (*t == '%' || *t == '\n')) { s = next_line(s, end); continue; }
        break;
    }
    if (s >= end) return fail(VBNMF_ERR_BAD_ARG, "%s: no size line", path);
    int64_t n = 0, m = 0, nent = 0;
    {
        const char *q = parse_uint(s, end, n);
        if (q) q = parse_uint(q, end, m);
        if (q && coordinate) q = parse_uint(q, end, nent);
        if (!q) return fail(VBNMF_ERR_BAD_ARG, "%s: malformed size line", path);
        s = next_line(q, end);
    }
    if (n <= 0 || m <= 0) return fail(VBNMF_ERR_BAD_ARG, "%s: matrix dimensions must be positive (got %lld x %lld)", path, (long long)n, (long long)m);
    if (n > 0x7FFFFFFFLL - 64 || m > 0x7FFFFFFFLL - 64) return fail(VBNMF_ERR_BAD_ARG, "%s: a matrix dimension exceeds 2^31-65", path);
    if (sym && n != m) return fail(VBNMF_ERR_BAD_ARG, "%s: a symmetric matrix must be square", path);

    std::vector<int32_t> rows, cols;
    std::vector<double> vals;
    if (!coordinate) {
        // dense array: column-major values, one per line (general: n*m of them; symmetric: lower triangle by columns)
        const int64_t expect = sym == 0 ? n * m : (sym == 1 ? n * (n + 1) / 2 : n * (n - 1) / 2);
        int64_t i = sym == 2 ? 1 : 0, j = 0, got = 0;
        while (s < end && got < expect) {
            const char *t = skip_space(s, end);
            if (t >= end) break;
            if (*t == '\n' || *t == '%') { s = next_line(s, end); continue; }
            double v;
            const char *q = parse_real(t, end, v);
            if (!q) return fail(VBNMF_ERR_BAD_ARG, "%s: malformed value at byte %lld", path, (long long)(t - f.p));
            if (v != 0.0) {
                rows.push_back((int32_t)i); cols.push_back((int32_t)j); vals.push_back(v);
                if (sym && i != j) { rows.push_back((int32_t)j); cols.push_back((int32_t)i); vals.push_back(sym == 2 ? -v : v); }
            }
            got++;
            if (++i == n) { j++; i = sym == 0 ? 0 : (sym == 1 ? j : j + 1); }
            s = next_line(q, end);
        }
        if (got != expect) return fail(VBNMF_ERR_BAD_ARG, "%s: %lld values found, %lld expected", path, (long long)got, (long long)expect);
    } else {
        if (nent > 0x7FFFFFFFLL / (sym ? 2 : 1)) return fail(VBNMF_ERR_BAD_ARG, "%s: too many entries (%lld)", path, (long long)nent);
        // cut the body at line starts, one piece per thread
        const int T = std::max(1, std::min(host_threads(), (int)((end - s) / (1 << 20)) + 1));
        std::vector<const char *> cut(T + 1);
        cut[0] = s; cut[T] = end;
        for (int t = 1; t < T; t++) {
            const char *c = s + (size_t)((end - s) / T) * t;
            cut[t] = c <= cut[t - 1] ? cut[t - 1] : next_line(c, end);
        }
        std::vector<Piece> pieces(T);
        parallel_for(T, [&](int64_t b, int64_t e, int) {
            for (int64_t t = b; t < e; t++) {
                Piece &P = pieces[t];
                const char *q = cut[t], *qe = cut[t + 1];
                const size_t guess = (size_t)(qe - q) / 8 + 16;
                P.row.reserve(guess); P.col.reserve(guess); P.val.reserve(guess);
                while (q < qe) {
                    const char *a = skip_space(q, qe);
                    if (a >= qe) break;
                    if (*a == '\n' || *a == '%') { q = next_line(a, qe); continue; }
                    int64_t i, j;
                    double v = 1.0;
                    const char *c = parse_uint(a, qe, i);
                    if (c) c = parse_uint(c, qe, j);
                    if (c && !pattern) c = parse_real(c, qe, v);
                    if (!c) { P.bad = 1; P.bad_at = a - f.p; return; }
                    if (i < 1 || i > n || j < 1 || j > m) { P.bad = 2; P.bad_at = a - f.p; return; }
                    P.row.push_back((int32_t)(i - 1)); P.col.push_back((int32_t)(j - 1)); P.val.push_back(v);
                    P.lines++;
                    q = next_line(c, qe);
                }
            }
        }, T);
        int64_t total = 0;
        for (const Piece &P : pieces) {
            if (P.bad == 1) return fail(VBNMF_ERR_BAD_ARG, "%s: malformed entry line at byte %lld", path, (long long)P.bad_at);
            if (P.bad == 2) return fail(VBNMF_ERR_BAD_ARG, "%s: entry at byte %lld is outside the %lld x %lld matrix", path, (long long)P.bad_at, (long long)n, (long long)m);
            total += P.lines;
        }
        if (total != nent) return fail(VBNMF_ERR_BAD_ARG, "%s: %lld entry lines found, the size line says %lld", path, (long long)total, (long long)nent);
        rows.reserve(total * (sym ? 2 : 1)); cols.reserve(total * (sym ? 2 : 1)); vals.reserve(total * (sym ? 2 : 1));
        for (Piece &P : pieces) {
            for (size_t q = 0; q < P.row.size(); q++) {
                rows.push_back(P.row[q]); cols.push_back(P.col[q]); vals.push_back(P.val[q]);
                if (sym && P.row[q] != P.col[q]) { rows.push_back(P.col[q]); cols.push_back(P.row[q]); vals.push_back(sym == 2 ? -P.val[q] : P.val[q]); }
            }
            std::vector<int32_t>().swap(P.row); std::vector<int32_t>().swap(P.col); std::vector<double>().swap(P.val);
        }
    }
    // counting sort by column (stable: file order inside a column), then the shared canonicaliser
    const int64_t nin = (int64_t)rows.size();
    std::vector<int32_t> p(m + 1, 0), ri(nin);
    std::vector<double> xv(nin);
    for (int64_t e = 0; e < nin; e++) p[cols[e] + 1]++;
    for (int64_t j = 0; j < m; j++) p[j + 1] += p[j];
    {
        std::vector<int32_t> cur(p.begin(), p.end() - 1);
        for (int64_t e = 0; e < nin; e++) { const int32_t o = cur[cols[e]]++; ri[o] = rows[e]; xv[o] = vals[e]; }
    }
    std::vector<int32_t>().swap(rows); std::vector<int32_t>().swap(cols); std::vector<double>().swap(vals);
    return matrix_from_csc(n, m, p.data(), ri.data(), xv.data(), X);
}

// What Matrix::writeMM writes for a general sparse matrix: banner, size line, then "i j value" by columns.
int matrix_write_mtx(const Matrix &X, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(VBNMF_ERR_BAD_ARG, "cannot open %s for writing (%s)", path, strerror(errno));
    bool integer = true;
    for (int64_t e = 0; e < X.nnz && integer; e++) integer = X.val[e] == std::floor(X.val[e]) && std::fabs(X.val[e]) < 9e15;
    fprintf(fp, "%%%%MatrixMarket matrix coordinate %s general\n", integer ? "integer" : "real");
    fprintf(fp, "%lld %lld %lld\n", (long long)X.n, (long long)X.m, (long long)X.nnz);
    for (int64_t j = 0; j < X.m; j++)
        for (int64_t e = X.colptr[j]; e < X.colptr[j + 1]; e++) {
            if (integer) fprintf(fp, "%d %lld %lld\n", X.row[e] + 1, (long long)(j + 1), (long long)X.val[e]);
            else fprintf(fp, "%d %lld %.17g\n", X.row[e] + 1, (long long)(j + 1), X.val[e]);
        }
    if (fclose(fp) != 0) return fail(VBNMF_ERR_BAD_ARG, "writing %s failed (%s)", path, strerror(errno));
    return VBNMF_OK;
}

}  // namespace vbnmf

using namespace vbnmf;

extern "C" {

int vbnmf_matrix_from_mtx(const char *path, vbnmf_matrix **out)
{
    if (!path) return fail(VBNMF_ERR_BAD_ARG, "path is NULL");
    return new_matrix(out, [&](Matrix &M) { return matrix_from_mtx(path, M); });
}

int vbnmf_matrix_write_mtx(const vbnmf_matrix *X, const char *path)
{
    if (!X || !path) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (X->M.shell) return fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries");
    return matrix_write_mtx(X->M, path);
}

int vbnmf_matrix_csc(const vbnmf_matrix *X, const int64_t **colptr, const int32_t **row, const double **val)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (X->M.shell) return fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries");
    if (colptr) *colptr = X->M.colptr.data();
    if (row) *row = X->M.row.data();
    if (val) *val = X->M.val.data();
    return VBNMF_OK;
}

}  // extern "C"
