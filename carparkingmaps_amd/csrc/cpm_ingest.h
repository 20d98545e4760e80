// cpm_ingest.h -- the data formats either side of the sampler path:
//   createdatamatrix   src/createdatamatrix.jl:3-27  Uber Movement CSV rows -> dense datamatrix[Z][Z][T][2]
//   processgeodata     src/processgeodata.jl:148-166 zone centroids -> distance_matrix_km[Z][Z]
// MI355X-first: the dense arrays (6.4 GB + 134 MB at Z = 4096) are only ever built in HBM.  The host
// parses the CSV text (memory-mapped, one slice of lines per thread) into five columns and hands
// them over once; the reference's row loop, with its "later rows overwrite earlier ones" rule,
// becomes two order-free passes over an owner array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace cpm {

// ---------------------------------------------------------------------------------------
// createdatamatrix, device side.  raw = rawdata[:,1:5] column-major (n x 5 Float64: sourceid,
// dstid, hod, mean_travel_time, standard_deviation_travel_time).  Zone id 0 -> Z on both
// endpoints, hod 0 -> 24 (:9-17); indices must be integers in range (the reference raises
// InexactError / BoundsError: here *err = 1 and the row is ignored).
// Pass 1: owner[cell] = max(row + 1) over the rows that address the cell -- the LAST such row,
// which is the one whose values survive the reference's sequential loop (:21-22).
// Pass 2: that row writes mean and std.  dm is zeroed by the caller (:7).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool dm_cell(const double *__restrict__ raw, int64_t n, int64_t r, int Z, int T, size_t &cell)
{
    const double a = raw[r], b = raw[n + r], h = raw[2 * n + r];
    if (!(a >= 0.0 && a <= Z && b >= 0.0 && b <= Z && h >= 0.0 && h <= T)) return false;
    if (a != floor(a) || b != floor(b) || h != floor(h)) return false;
    int i1 = static_cast<int>(a), i2 = static_cast<int>(b), i3 = static_cast<int>(h);
    if (i1 == 0) i1 = Z;   // :9-11
    if (i2 == 0) i2 = Z;   // :12-14
    if (i3 == 0) i3 = 24;  // :15-17 (the literal 24 of the reference, not T)
    if (i3 > T) return false;
    cell = static_cast<size_t>(i1 - 1) + static_cast<size_t>(Z) * (static_cast<size_t>(i2 - 1) + static_cast<size_t>(Z) * (i3 - 1));
    return true;
}

__global__ __launch_bounds__(256) void k_dm_owner(const double *__restrict__ raw, int64_t n, int Z, int T,
                                                  uint32_t *__restrict__ owner, int *__restrict__ err)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    size_t cell;
    if (!dm_cell(raw, n, r, Z, T, cell)) {
        *err = 1;
        return;
    }
    atomicMax(&owner[cell], static_cast<uint32_t>(r + 1));
}

__global__ __launch_bounds__(256) void k_dm_write(const double *__restrict__ raw, int64_t n, int Z, int T,
                                                  const uint32_t *__restrict__ owner, double *__restrict__ dm)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    size_t cell;
    if (!dm_cell(raw, n, r, Z, T, cell)) return;
    if (owner[cell] != static_cast<uint32_t>(r + 1)) return;
    dm[cell] = raw[3 * n + r];                                      // mean_travel_time (:21)
    dm[cell + static_cast<size_t>(Z) * Z * T] = raw[4 * n + r];     // standard_deviation_travel_time (:22)
}

// ---------------------------------------------------------------------------------------
// processgeodata, distance part (src/processgeodata.jl:148-166): for i != j
//   111.3 * sqrt(cos((lat_i + lat_j)/2 * 0.01745)^2 * (long_i - long_j)^2 + (lat_i - lat_j)^2),
// diagonal 1 km.  One thread per cell; coalesced stores.  (cos is the device library's: it agrees
// with glibc / Julia to an ulp or two, not bit for bit; everything else is plain IEEE arithmetic.)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_distance(const double *__restrict__ lat, const double *__restrict__ lon, int Z,
                                                  double *__restrict__ dist)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= Z) return;
    double d = 1.0;
    if (i != j) {
        const double c_lat_long = 111.3, conv_deg_rad = 0.01745;
        const double lat1 = lat[i], lat2 = lat[j], long1 = lon[i], long2 = lon[j];
        const double cs = cos((lat1 + lat2) / 2 * conv_deg_rad);
        const double dl = long1 - long2, dp = lat1 - lat2;
        d = c_lat_long * sqrt((cs * cs) * (dl * dl) + dp * dp);
    }
    dist[static_cast<size_t>(j) * Z + i] = d;
}

// ---------------------------------------------------------------------------------------
// Uber Movement CSV text -> five Float64 columns (host).  Header line skipped; the first five
// comma-separated fields of every line are parsed (sourceid,dstid,hod,mean_travel_time,
// standard_deviation_travel_time; further fields ignored, src/createdatamatrix.jl:5); blank
// lines are skipped; "\r\n" accepted.  The file is memory-mapped and cut at line ends into one
// slice per thread; slices are parsed independently and concatenated in file order, so row r
// of the result is line r + 2 of the file (the order the last-row-wins rule depends on).
// ---------------------------------------------------------------------------------------
struct CsvRows {
    std::vector<double> col[5];
    int64_t n = 0;
};

// One numeric field.  Fast path (Clinger): up to 15 significant digits and at most 22 decimals, no exponent -> the value
// is mantissa / 10^decimals with both operands exact doubles, so the one IEEE division is correctly rounded: the same
// double strtod (and Julia's CSV parser) produce.  Anything else (exponents, long digit strings, inf/nan) goes to strtod.
inline bool parse_field(const char *q, const char *le, double &v, const char *&stop)
{
    static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *p = q;
    while (p < le && (*p == ' ' || *p == '\t')) ++p;
    const char *s0 = p;
    bool neg = false;
    if (p < le && (*p == '-' || *p == '+')) neg = (*p++ == '-');
    uint64_t m = 0;
    int digits = 0, decimals = 0;
    bool any = false, slow = false;
    while (p < le && *p >= '0' && *p <= '9') {
        if (m || *p != '0') ++digits;
        if (digits <= 15) m = m * 10 + static_cast<uint64_t>(*p - '0');
        else slow = true;
        any = true;
        ++p;
    }
    if (p < le && *p == '.') {
        ++p;
        while (p < le && *p >= '0' && *p <= '9') {
            if (m || *p != '0') ++digits;
            if (digits <= 15) m = m * 10 + static_cast<uint64_t>(*p - '0');
            else slow = true;
            ++decimals;
            any = true;
            ++p;
        }
    }
    if (any && !slow && decimals <= 22 && !(p < le && (*p == 'e' || *p == 'E'))) {
        v = static_cast<double>(m) / p10[decimals];
        if (neg) v = -v;
        stop = p;
        return true;
    }
    char *e = nullptr;
    v = strtod(s0, &e);  // never reads past the line: a number holds no '\n', and the mapping ends in '\n' or in a guard byte
    stop = e;
    return e != s0 && e <= le;
}

inline bool parse_csv_slice(const char *p, const char *end, CsvRows &out, std::string &err, int64_t &bad_line)
{
    int64_t line = 0;
    while (p < end) {
        const char *eol = static_cast<const char *>(memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *le = eol ? eol : end;
        const char *q = p;
        while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if (q < le) {
            double v[5];
            for (int k = 0; k < 5; ++k) {
                const char *stop = nullptr;
                if (!parse_field(q, le, v[k], stop)) {
                    err = "malformed field " + std::to_string(k + 1);
                    bad_line = line;
                    return false;
                }
                q = stop;
                while (q < le && (*q == ' ' || *q == '\t')) ++q;
                if (k < 4) {
                    if (q >= le || *q != ',') {
                        err = "fewer than five fields";
                        bad_line = line;
                        return false;
                    }
                    ++q;
                }
            }
            for (int k = 0; k < 5; ++k) out.col[k].push_back(v[k]);
            ++out.n;
        }
        ++line;
        p = eol ? eol + 1 : end;
    }
    return true;
}

// returns "" on success
inline std::string read_uber_csv(const char *path, int threads, CsvRows &rows, double *seconds_out)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return std::string("open ") + path + ": " + strerror(errno);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return std::string("stat ") + path + ": " + strerror(errno);
    }
    const size_t size = static_cast<size_t>(st.st_size);
    if (size == 0) {
        close(fd);
        rows = CsvRows();
        return "";
    }
    // one guard byte beyond the file keeps strtod inside the mapping when the last line has no '\n':
    // the kernel zero-fills the rest of the last page; a file that ends exactly on a page boundary is copied instead
    const long page = sysconf(_SC_PAGESIZE);
    const char *data = nullptr;
    void *map = MAP_FAILED;
    std::vector<char> copy;
    if (size % static_cast<size_t>(page) != 0) {
        map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map == MAP_FAILED) {
            close(fd);
            return std::string("mmap ") + path + ": " + strerror(errno);
        }
        data = static_cast<const char *>(map);
    } else {
        copy.resize(size + 1);
        size_t got = 0;
        while (got < size) {
            const ssize_t k = pread(fd, copy.data() + got, size - got, static_cast<off_t>(got));
            if (k <= 0) {
                close(fd);
                return std::string("read ") + path + ": " + strerror(errno);
            }
            got += static_cast<size_t>(k);
        }
        copy[size] = '\0';
        data = copy.data();
    }
    close(fd);
    const char *end = data + size;
    const char *body = static_cast<const char *>(memchr(data, '\n', size));  // header line (:4: CSV.read takes it as column names)
    body = body ? body + 1 : end;
    const size_t blen = static_cast<size_t>(end - body);
    int nt = std::max(1, std::min<int>(threads, static_cast<int>(blen / (1 << 20)) + 1));
    std::vector<const char *> cut(nt + 1);
    cut[0] = body;
    cut[nt] = end;
    for (int k = 1; k < nt; ++k) {
        const char *c = body + blen * k / nt;
        const char *nl = static_cast<const char *>(memchr(c, '\n', static_cast<size_t>(end - c)));
        cut[k] = nl ? nl + 1 : end;
    }
    std::vector<CsvRows> part(nt);
    std::vector<std::string> perr(nt);
    std::vector<int64_t> pbad(nt, -1);
    std::vector<std::thread> th;
    for (int k = 0; k < nt; ++k)
        th.emplace_back([&, k] {
            if (cut[k] < cut[k + 1]) parse_csv_slice(cut[k], cut[k + 1], part[k], perr[k], pbad[k]);
        });
    for (auto &t : th) t.join();
    std::string err;
    int64_t lines_before = 0;
    for (int k = 0; k < nt && err.empty(); ++k) {
        if (!perr[k].empty()) err = std::string(path) + ": line " + std::to_string(lines_before + pbad[k] + 2) + ": " + perr[k];
        for (const char *c = cut[k]; c < cut[k + 1]; ++c) lines_before += (*c == '\n');
    }
    if (map != MAP_FAILED) munmap(map, size);
    if (!err.empty()) return err;
    int64_t n = 0;
    for (auto &p : part) n += p.n;
    rows.n = n;
    for (int c = 0; c < 5; ++c) {
        rows.col[c].clear();
        rows.col[c].reserve(static_cast<size_t>(n));
        for (auto &p : part) rows.col[c].insert(rows.col[c].end(), p.col[c].begin(), p.col[c].end());
    }
    (void)seconds_out;
    return "";
}

}  // namespace cpm
