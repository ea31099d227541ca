// pcl_io.hip — host-side reader of the datasets' text point clouds (data_utils.py:16-43 read_stanford, :138-163
// read_omniscenes: `pandas.read_table(path, header=None, delim_whitespace=True).values`, one point per line,
// "x y z r g b").  No device code: the file is mmap-ed, cut into line-aligned slices and parsed by a pool of threads
// straight into the caller's row-major double array (the layout `.values` has), ready for one upload.
//
// Number parsing restates the default float converter of pandas' C engine (see parse_number), so the table holds the very
// doubles the reference gets.
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/piccolo_hip.h"

namespace {

struct Mapped {
    const char* p = nullptr;
    size_t len = 0;
    int fd = -1;
    int open(const char* path)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return -errno;
        struct stat st;
        if (fstat(fd, &st) != 0) { int e = errno; ::close(fd); fd = -1; return -e; }
        len = (size_t)st.st_size;
        if (len == 0) return 0;
        void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);    // one pass of page-table set-up, not a fault per 4 KiB
        if (m == MAP_FAILED) { int e = errno; ::close(fd); fd = -1; return -e; }
        (void)madvise(m, len, MADV_SEQUENTIAL);
        p = (const char*)m;
        return 0;
    }
    ~Mapped()
    {
        if (p) munmap((void*)p, len);
        if (fd >= 0) ::close(fd);
    }
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// a line counts when it holds anything but blanks (pandas: skip_blank_lines=True)
int64_t count_rows(const char* b, const char* e)
{
    int64_t rows = 0;
    bool content = false;
    for (const char* q = b; q < e; q++) {
        if (*q == '\n') { rows += content; content = false; }
        else if (!is_blank(*q)) content = true;
    }
    return rows + content;
}

// 1e0 .. 1e308 as correctly rounded double literals (the table pandas' converter scales with)
#define P10_ROW(a) 1e##a##0, 1e##a##1, 1e##a##2, 1e##a##3, 1e##a##4, 1e##a##5, 1e##a##6, 1e##a##7, 1e##a##8, 1e##a##9
const double kPow10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9,
                         P10_ROW(1), P10_ROW(2), P10_ROW(3), P10_ROW(4), P10_ROW(5), P10_ROW(6), P10_ROW(7), P10_ROW(8), P10_ROW(9),
                         P10_ROW(10), P10_ROW(11), P10_ROW(12), P10_ROW(13), P10_ROW(14), P10_ROW(15), P10_ROW(16), P10_ROW(17),
                         P10_ROW(18), P10_ROW(19), P10_ROW(20), P10_ROW(21), P10_ROW(22), P10_ROW(23), P10_ROW(24), P10_ROW(25),
                         P10_ROW(26), P10_ROW(27), P10_ROW(28), P10_ROW(29),
                         1e300, 1e301, 1e302, 1e303, 1e304, 1e305, 1e306, 1e307, 1e308};
#undef P10_ROW

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// Parses one field starting at q (no leading blanks) and ending before `end`; returns the first unread character or
// nullptr when the field is not a number.
//
// The conversion restates the default ("high" precision) float converter of pandas' C parser — third-party:
// pandas/_libs/src/parser/tokenizer.c, precise_xstrtod; `pandas` is unpinned in requirements.txt:11, the goldens were
// produced with the pandas of the build image — because that is what the reference's arrays hold: the first 17 digits
// (leading zeros included) are accumulated in a double, the rest only shift the decimal exponent, and the result is
// scaled by ONE multiplication or division with a tabulated power of ten.  Exact for the datasets' short decimals; up
// to an ulp away from strtod for 16-17 digit fields, where matching the reference bit for bit needs this algorithm.
const char* parse_number(const char* q, const char* end, double* out)
{
    const char* start = q;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    const int max_digits = 17;
    double number = 0.;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    while (q < end && is_digit(*q)) {
        if (num_digits < max_digits) { number = number * 10. + (*q - '0'); num_digits++; }
        else ++exponent;
        q++;
    }
    if (q < end && *q == '.') {
        q++;
        while (num_digits < max_digits && q < end && is_digit(*q)) {
            number = number * 10. + (*q - '0');
            q++; num_digits++; num_decimals++;
        }
        if (num_digits >= max_digits)
            while (q < end && is_digit(*q)) ++q;                       // extra decimals are dropped
        exponent -= num_decimals;
    }
    if (num_digits > 0) {
        if (neg) number = -number;
        if (q < end && (*q == 'e' || *q == 'E')) {
            const char* r = q + 1;
            bool eneg = false;
            if (r < end && (*r == '-' || *r == '+')) { eneg = *r == '-'; r++; }
            if (r < end && is_digit(*r)) {
                int n = 0;
                while (r < end && is_digit(*r)) { if (n < 100000) n = n * 10 + (*r - '0'); r++; }
                exponent += eneg ? -n : n;
                q = r;
            }                                                           // no digits after 'e': it is not consumed
        }
        if (!(q == end || is_blank(*q))) return nullptr;
        if (exponent > 308) return nullptr;                            // out of range: the reference's column would not be numeric
        if (exponent > 0) number *= kPow10[exponent];
        else if (exponent < -308) {
            if (exponent < -616) number = 0.;
            else { number /= kPow10[-308 - exponent]; number /= kPow10[308]; }
        } else number /= kPow10[-exponent];
        if (std::isinf(number)) return nullptr;
        *out = number;
        return q;
    }
    // no digits: the nan / inf spellings (copies the field: the mapping is not NUL terminated)
    const char* fe = start;
    while (fe < end && !is_blank(*fe)) fe++;
    size_t flen = (size_t)(fe - start);
    if (flen == 0 || flen > 16) return nullptr;
    char buf[17];
    memcpy(buf, start, flen);
    buf[flen] = 0;
    char* stop = nullptr;
    double v = strtod(buf, &stop);
    if (stop != buf + flen || std::isfinite(v)) return nullptr;
    *out = v;
    return fe;
}

// Parses the rows of [b, e) into out (row-major, `cols` per row).  Returns 0, or the 1-based line (relative to b) of the
// first malformed row as a negative number.
int64_t parse_slice(const char* b, const char* e, int cols, double* out)
{
    int64_t line = 0;
    const char* q = b;
    while (q < e) {
        line++;
        const char* eol = (const char*)memchr(q, '\n', (size_t)(e - q));
        if (!eol) eol = e;
        const char* r = q;
        while (r < eol && is_blank(*r)) r++;
        if (r < eol) {
            for (int c = 0; c < cols; c++) {
                while (r < eol && is_blank(*r)) r++;
                if (r >= eol) return -line;                              // too few fields
                r = parse_number(r, eol, out + c);
                if (!r) return -line;
            }
            while (r < eol && is_blank(*r)) r++;
            if (r < eol) return -line;                                   // too many fields
            out += cols;
        }
        q = eol + 1;
    }
    return 0;
}

}  // namespace

// Rows (non-blank lines) of a whitespace-separated text file; < 0: -errno.
extern "C" int64_t pcl_cloud_txt_rows(const char* path)
{
    if (!path) return -EINVAL;
    Mapped m;
    int rc = m.open(path);
    if (rc) return rc;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (m.len < (1u << 20)) nt = 1;
    std::vector<int64_t> part(nt, 0);
    std::vector<const char*> cut(nt + 1);
    cut[0] = m.p; cut[nt] = m.p + m.len;
    for (unsigned t = 1; t < nt; t++) {
        const char* q = m.p + m.len / nt * t;
        const char* nl = (const char*)memchr(q, '\n', (size_t)(m.p + m.len - q));
        cut[t] = nl ? nl + 1 : m.p + m.len;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) pool.emplace_back([&, t] { part[t] = cut[t] < cut[t + 1] ? count_rows(cut[t], cut[t + 1]) : 0; });
    for (auto& th : pool) th.join();
    int64_t rows = 0;
    for (auto v : part) rows += v;
    return rows;
}

// Parses `rows` x `cols` numbers into out.  Returns 0; -errno for IO errors; PCL_EINVAL for bad arguments or a row count
// that does not match; otherwise -(1000 + line) for the first malformed line (1-based; fields missing, extra or not numeric).
extern "C" int64_t pcl_cloud_txt_read(const char* path, int64_t rows, int cols, double* out, int nthreads)
{
    if (!path || !out || rows < 0 || cols <= 0) return PCL_EINVAL;
    Mapped m;
    int rc = m.open(path);
    if (rc) return rc;
    unsigned nt = nthreads > 0 ? (unsigned)nthreads : std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (m.len < (1u << 20)) nt = 1;
    std::vector<const char*> cut(nt + 1);
    cut[0] = m.p; cut[nt] = m.p + m.len;
    for (unsigned t = 1; t < nt; t++) {
        const char* q = m.p + m.len / nt * t;
        const char* nl = (const char*)memchr(q, '\n', (size_t)(m.p + m.len - q));
        cut[t] = nl ? nl + 1 : m.p + m.len;
    }
    // rows and physical lines before every slice, so that each thread knows where to write and errors name the line
    std::vector<int64_t> first_row(nt + 1, 0), first_line(nt + 1, 0);
    {
        std::vector<std::thread> pool;
        std::vector<int64_t> r(nt, 0), l(nt, 0);
        for (unsigned t = 0; t < nt; t++)
            pool.emplace_back([&, t] {
                if (cut[t] >= cut[t + 1]) return;
                r[t] = count_rows(cut[t], cut[t + 1]);
                int64_t lines = 0;
                for (const char* q = cut[t]; q < cut[t + 1]; q++) lines += *q == '\n';
                l[t] = lines;
            });
        for (auto& th : pool) th.join();
        for (unsigned t = 0; t < nt; t++) { first_row[t + 1] = first_row[t] + r[t]; first_line[t + 1] = first_line[t] + l[t]; }
    }
    if (first_row[nt] != rows) return PCL_EINVAL;
#ifdef MADV_POPULATE_WRITE
    {   // fault the (freshly allocated) output in with one call: eight threads taking first-touch faults on one address
        // space serialise on the mm lock and cost more than the parse itself
        uintptr_t lo = ((uintptr_t)out + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)(out + rows * cols)) & ~(uintptr_t)4095;
        if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE);
    }
#endif
    std::vector<int64_t> err(nt, 0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++)
        pool.emplace_back([&, t] {
            if (cut[t] < cut[t + 1]) err[t] = parse_slice(cut[t], cut[t + 1], cols, out + first_row[t] * cols);
        });
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < nt; t++)
        if (err[t] < 0) return -(1000 + first_line[t] + (-err[t]));
    return 0;
}
