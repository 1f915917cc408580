// fasta.cpp — part of libmprime_hip.so: native FASTA record parser behind include/mprime_host.h (H1).
// Restates the record semantics of parse_seq (scripts/multiPrime-core_V20.py:441-455) on raw bytes, on several
// threads: the file is read with parallel pread()s, cut into chunks at line starts, every chunk is scanned for
// lines (terminators \n, \r\n, \r — Python's universal newlines), and one serial pass joins the chunks: it keeps
// the ids in first-appearance order and makes a repeated id continue its first record (defaultdict(str)).
// The per-character mapping of V20:453 is NOT done here — that is device work (pack_kernel, mp_load_msa).
#include "../../include/mprime.h"
#include "workers.hpp"
#include "../../include/mprime_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <chrono>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Seg { int64_t off; int32_t len; int32_t row; };          // one stripped sequence line
struct Event {                                                   // per line that matters, in file order
    int64_t off;
    int32_t len;
    int32_t header;             // 1: id token [off, off+len); 0: sequence line
    uint64_t hash;              // of the id token
};

// str.strip() of a text-mode line: every character str.isspace() is true for.  ASCII: space, \t \n \v \f \r and the separators
// 0x1c-0x1f (bytes.strip() would leave those); beyond it ([r4], recorded from the reference's parse_seq: tests/golden/parser_ws.json)
// U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000 in their UTF-8 forms (the reference reads text mode, UTF-8).
inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
inline bool is_space2(uint8_t a, uint8_t b) { return a == 0xC2 && (b == 0x85 || b == 0xA0); }
inline bool is_space3(uint8_t a, uint8_t b, uint8_t c) {
    if (a == 0xE1) return b == 0x9A && c == 0x80;                                             // U+1680
    if (a == 0xE2) return (b == 0x80 && ((c >= 0x80 && c <= 0x8A) || c == 0xA8 || c == 0xA9 || c == 0xAF)) || (b == 0x81 && c == 0x9F);
    if (a == 0xE3) return b == 0x80 && c == 0x80;                                             // U+3000
    return false;
}
// [a, z) without the white space at either end
inline void strip_line(const uint8_t *b, int64_t &a, int64_t &z) {
    for (;;) {
        if (a < z && is_space(b[a])) a += 1;
        else if (a + 1 < z && is_space2(b[a], b[a + 1])) a += 2;
        else if (a + 2 < z && is_space3(b[a], b[a + 1], b[a + 2])) a += 3;
        else break;
    }
    for (;;) {
        if (z > a && is_space(b[z - 1])) z -= 1;
        else if (z - 1 > a && is_space2(b[z - 2], b[z - 1])) z -= 2;
        else if (z - 2 > a && is_space3(b[z - 3], b[z - 2], b[z - 1])) z -= 3;
        else break;
    }
}

inline uint64_t hash_bytes(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ULL ^ (n * 0x9E3779B97F4A7C15ULL);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, p + i, 8);
        h = (h ^ x) * 0x100000001b3ULL;
        h ^= h >> 29;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ULL;
    h ^= h >> 32; h *= 0xff51afd7ed558ccdULL; h ^= h >> 29;
    return h;
}

}  // namespace

struct mp_fasta {
    char err[512] = {0};
    uint8_t *own = nullptr;           // file contents when parsed from a path (big_alloc)
    size_t own_bytes = 0;
    const uint8_t *buf = nullptr;
    int64_t n = 0;
    int n_threads = 1;
    std::vector<int64_t> id_off_src;  // per row: offset of the id token in buf
    std::vector<int32_t> id_len;
    std::vector<Seg> segs;            // grouped by row, file order inside a row
    std::vector<int64_t> row_seg;     // [n_rows+1] segment range of each row
    std::vector<int64_t> row_off;     // [n_rows+1] residue offsets
    int64_t id_bytes = 0;
    ~mp_fasta();
};

namespace {

// Buffers of file size: anonymous mappings advised to use huge pages — first touch of a gigabyte costs ~500 page faults instead
// of ~260 000 (the parallel pread()s of a large file spend most of their time in those faults otherwise).
uint8_t *big_alloc(size_t n, size_t *mapped, bool want_huge) {
    const size_t huge = (size_t)2 << 20;
    const size_t len = (n + huge) / huge * huge;
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
#ifdef MADV_HUGEPAGE
    if (want_huge && len >= 2 * huge) (void)madvise(p, len, MADV_HUGEPAGE);
#endif
    *mapped = len;
    return (uint8_t *)p;
}

int ffail(mp_fasta *f, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(f->err, sizeof f->err, fmt, ap);
    va_end(ap);
    return code;
}

// MP_HOST_TRACE=1: stage times of the parser on stderr
struct Trace {
    bool on = getenv("MP_HOST_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mprime host] %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int threads_for(int asked, int64_t bytes) {
    if (const char *e = getenv("MP_HOST_THREADS")) return std::max(1, atoi(e));     // exact (tests cut tiny inputs into chunks)
    int n = asked;
    if (n <= 0) {
        n = (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 1;
        n = std::min(n, 32);
    }
    int64_t by_size = bytes / (1 << 20) + 1;              // no point in a thread per few kilobytes
    if ((int64_t)n > by_size) n = (int)by_size;
    return std::max(n, 1);
}

// first line start at or after `pos`
int64_t line_start_at_or_after(const uint8_t *b, int64_t n, int64_t pos) {
    if (pos <= 0) return 0;
    if (pos >= n) return n;
    uint8_t prev = b[pos - 1];
    if (prev == '\n') return pos;
    if (prev == '\r') return b[pos] == '\n' ? pos + 1 : pos;
    for (int64_t i = pos; i < n; i++) {
        if (b[i] == '\n') return i + 1;
        if (b[i] == '\r') return (i + 1 < n && b[i + 1] == '\n') ? i + 2 : i + 1;
    }
    return n;
}

void scan_chunk(const uint8_t *b, int64_t n, int64_t begin, int64_t end, std::vector<Event> &out) {
    int64_t p = begin;
    while (p < end) {
        // the line is [p, q), q = its terminator or the end of the buffer
        int64_t q = p;
        {
            // \n or a lone \r, whichever comes first.  The search window doubles until it holds one of them, so a file whose
            // lines end in \r alone is not scanned to its end once per line.
            int64_t lim = std::min<int64_t>(n, p + 256);
            for (;;) {
                const uint8_t *nl = (const uint8_t *)memchr(b + p, '\n', (size_t)(lim - p));
                const int64_t qn = nl ? (int64_t)(nl - b) : lim;
                const uint8_t *cr = (const uint8_t *)memchr(b + p, '\r', (size_t)(qn - p));
                q = cr ? (int64_t)(cr - b) : qn;
                if (nl || cr || lim == n) break;
                lim = std::min<int64_t>(n, p + 2 * (lim - p));
            }
        }
        int64_t next = q;
        if (q < n) next = (b[q] == '\r' && q + 1 < n && b[q + 1] == '\n') ? q + 2 : q + 1;
        if (q > p && b[p] == '#') { p = next; continue; }
        int64_t a = p, z = q;
        strip_line(b, a, z);
        if (q > p && b[p] == '>') {
            // i.strip().split(" ")[0]: the stripped line up to its first blank
            int64_t t = a;
            while (t < z && b[t] != ' ') t++;
            out.push_back(Event{a, (int32_t)(t - a), 1, hash_bytes(b + a, (size_t)(t - a))});
        } else {
            out.push_back(Event{a, (int32_t)(z - a), 0, 0});
        }
        p = next;
    }
}

// [r6] The join on all threads, for the file every real alignment is: no id occurs twice.  The serial join below walks 2 x 10^6 events
// of a 10^6-record file through a hash table in 19 ms — more than the scan of the gigabyte in front of it (6-9 ms on 32 threads).  Here
// every chunk of events (cut at line starts by the scan) takes the lines in front of the NEXT chunk's first header as the tail of its own
// last record, counts its records and segments, fills them at their prefix-sum positions, and all records go through one lock-free hash
// set (compare-and-swap on a record index per slot) that only has to answer one question: is any id repeated?  If one is — or the file
// starts with sequence data, or a chunk holds no header at all (a record longer than a chunk) — nothing is kept and the serial join
// decides, with its first-appearance order and its error message.  Records come out in file order, which IS first-appearance order
// when no id repeats; a header that receives no line creates no record (V20:454), as there.
bool join_parallel(mp_fasta *f, const std::vector<std::vector<Event>> &ev_all, std::vector<Seg> &raw) {
    if (getenv("MP_HOST_SERIAL_JOIN")) return false;
    std::vector<const std::vector<Event> *> ev;
    for (const auto &c : ev_all) if (!c.empty()) ev.push_back(&c);
    const int T = (int)ev.size();
    if (T < 2) return false;
    std::vector<size_t> fh((size_t)T + 1, 0);          // index of the first header of a chunk; fh[T] = 0: nothing follows the last chunk
    for (int t = 0; t < T; t++) {
        const auto &c = *ev[(size_t)t];
        size_t i = 0;
        while (i < c.size() && !c[i].header) i++;
        if (i == c.size()) return false;                // a chunk without a header
        fh[(size_t)t] = i;
    }
    if (fh[0] != 0) return false;                       // sequence data before the first header: the serial join reports it
    // the logical event run of chunk t: its own events from its first header on, then the next chunk's events before ITS first header
    auto walk = [&](int t, auto &&on_record, auto &&on_segment) {
        const auto &own = *ev[(size_t)t];
        const std::vector<Event> *nxt = t + 1 < T ? ev[(size_t)t + 1] : nullptr;
        const size_t n_own = own.size(), n_tail = nxt ? fh[(size_t)t + 1] : 0;
        bool open = false;                              // the pending header has received a line: its record exists
        const Event *pending = nullptr;
        for (size_t i = fh[(size_t)t]; i < n_own + n_tail; i++) {
            const Event &e = i < n_own ? own[i] : (*nxt)[i - n_own];
            if (e.header) { pending = &e; open = false; continue; }
            if (!open) { on_record(*pending); open = true; }
            if (e.len) on_segment(e);
        }
    };
    std::vector<int64_t> n_rec((size_t)T + 1, 0), n_seg((size_t)T + 1, 0);
    mp::run_on_threads(T, [&](int t) {
        int64_t r = 0, g = 0;
        walk(t, [&](const Event &) { r++; }, [&](const Event &) { g++; });
        n_rec[(size_t)t + 1] = r; n_seg[(size_t)t + 1] = g;
    });
    for (int t = 0; t < T; t++) { n_rec[(size_t)t + 1] += n_rec[(size_t)t]; n_seg[(size_t)t + 1] += n_seg[(size_t)t]; }
    const size_t R = (size_t)n_rec[(size_t)T], S = (size_t)n_seg[(size_t)T];
    if (R > 0x7fffffffULL - 1) return false;
    f->id_off_src.assign(R, 0);
    f->id_len.assign(R, 0);
    std::vector<uint64_t> id_hash(R);
    raw.assign(S, Seg{0, 0, 0});
    // ... with each record's first segment (row_seg) and, chunk by chunk, the residues before it (row_off: completed below)
    f->row_seg.assign(R + 1, 0);
    f->row_off.assign(R + 1, 0);
    std::vector<int64_t> chunk_bytes((size_t)T + 1, 0);
    mp::run_on_threads(T, [&](int t) {
        int64_t r = n_rec[(size_t)t] - 1, g = n_seg[(size_t)t], bytes = 0;
        walk(t, [&](const Event &h) {
                 r++;
                 f->id_off_src[(size_t)r] = h.off; f->id_len[(size_t)r] = h.len; id_hash[(size_t)r] = h.hash;
                 f->row_seg[(size_t)r] = g; f->row_off[(size_t)r] = bytes;
             },
             [&](const Event &e) { raw[(size_t)g++] = Seg{e.off, e.len, (int32_t)r}; bytes += e.len; });
        chunk_bytes[(size_t)t + 1] = bytes;
    });
    for (int t = 0; t < T; t++) chunk_bytes[(size_t)t + 1] += chunk_bytes[(size_t)t];
    f->row_seg[R] = (int64_t)S;
    f->row_off[R] = chunk_bytes[(size_t)T];
    mp::run_on_threads(T, [&](int t) {
        const int64_t add = chunk_bytes[(size_t)t];
        if (add) for (int64_t r = n_rec[(size_t)t]; r < n_rec[(size_t)t + 1]; r++) f->row_off[(size_t)r] += add;
    });
    // is any id repeated?
    size_t cap = 16;
    while (cap < 2 * R + 8) cap <<= 1;
    const size_t mask = cap - 1;
    std::unique_ptr<std::atomic<int32_t>[]> table(new (std::nothrow) std::atomic<int32_t>[cap]);
    if (!table) return false;
    mp::run_on_threads(T, [&](int t) { for (size_t i = cap * (size_t)t / (size_t)T, i1 = cap * ((size_t)t + 1) / (size_t)T; i < i1; i++) table[i].store(-1, std::memory_order_relaxed); });
    std::atomic<bool> repeated{false};
    const uint8_t *b = f->buf;
    mp::run_on_threads(T, [&](int t) {
        for (int64_t r = n_rec[(size_t)t]; r < n_rec[(size_t)t + 1] && !repeated.load(std::memory_order_relaxed); r++) {
            size_t h = (size_t)id_hash[(size_t)r] & mask;
            for (;;) {
                int32_t seen = -1;
                if (table[h].compare_exchange_strong(seen, (int32_t)r, std::memory_order_relaxed)) break;
                if (id_hash[(size_t)seen] == id_hash[(size_t)r] && f->id_len[(size_t)seen] == f->id_len[(size_t)r] &&
                    memcmp(b + f->id_off_src[(size_t)seen], b + f->id_off_src[(size_t)r], (size_t)f->id_len[(size_t)r]) == 0) {
                    repeated.store(true, std::memory_order_relaxed);
                    break;
                }
                h = (h + 1) & mask;
            }
        }
    });
    if (repeated.load()) {
        f->id_off_src.clear(); f->id_len.clear(); f->row_seg.clear(); f->row_off.clear(); raw.clear();
        return false;
    }
    f->id_bytes = 0;
    for (size_t r = 0; r < R; r++) f->id_bytes += f->id_len[r];
    return true;
}

int parse(mp_fasta *f) {
    const uint8_t *b = f->buf;
    const int64_t n = f->n;
    const int T = f->n_threads;
    std::vector<int64_t> cut((size_t)T + 1);
    for (int t = 0; t <= T; t++) cut[(size_t)t] = line_start_at_or_after(b, n, n * t / T);
    cut[(size_t)T] = n;
    std::vector<std::vector<Event>> ev((size_t)T);
    Trace tr;
    if (T == 1) scan_chunk(b, n, 0, n, ev[0]);
    else {
        mp::run_on_threads(T, [&](int t) { scan_chunk(b, n, cut[(size_t)t], cut[(size_t)t + 1], ev[(size_t)t]); });
    }
    tr.lap("scan");
    std::vector<Seg> raw;
    const bool fast = join_parallel(f, ev, raw);
    if (!fast) {
        // serial join: ids in first-appearance order, a repeated id continues its first record
        size_t n_ev = 0;
        for (auto &e : ev) n_ev += e.size();
        size_t cap = 16;
        while (cap < n_ev + 8) cap <<= 1;
        std::vector<int32_t> table(cap, -1);
        const size_t mask = cap - 1;
        raw.reserve(n_ev);
        std::vector<uint64_t> id_hash;
        // A record comes into being when its id RECEIVES a line (seq_dict[acc_id] += ..., V20:454) — a header that is
        // followed by another header creates nothing — and it takes its place in the order at that moment.
        int32_t cur = -1;
        bool have_header = false, resolved = false;
        Event pending{};
        for (auto &chunk : ev) {
            const size_t n_chunk = chunk.size();
            for (size_t ei = 0; ei < n_chunk; ei++) {
                const Event &e = chunk[ei];
                // the table slot of a header a few lines ahead is fetched while this line is handled (the probe is the one random
                // memory access of the loop)
                if (ei + 8 < n_chunk && chunk[ei + 8].header) __builtin_prefetch(&table[(size_t)chunk[ei + 8].hash & mask]);
                if (e.header) { pending = e; have_header = true; resolved = false; continue; }
                if (!have_header) return ffail(f, MP_ERR_ARG, "sequence data before the first '>' header");
                if (!resolved) {
                    size_t h = (size_t)pending.hash & mask;
                    for (;;) {
                        int32_t r = table[h];
                        if (r < 0) {
                            r = (int32_t)f->id_off_src.size();
                            table[h] = r;
                            f->id_off_src.push_back(pending.off);
                            f->id_len.push_back(pending.len);
                            id_hash.push_back(pending.hash);
                            cur = r;
                            break;
                        }
                        if (id_hash[(size_t)r] == pending.hash && f->id_len[(size_t)r] == pending.len &&
                            memcmp(b + f->id_off_src[(size_t)r], b + pending.off, (size_t)pending.len) == 0) { cur = r; break; }
                        h = (h + 1) & mask;
                    }
                    resolved = true;
                }
                if (e.len) raw.push_back(Seg{e.off, e.len, cur});
            }
            std::vector<Event>().swap(chunk);
        }
    }
    tr.lap("join");
    const size_t R = f->id_off_src.size();
    if (fast) {
        f->segs.swap(raw);                              // in file order = grouped by row; row_seg / row_off / id_bytes came with the join
    } else {
        // group the segments by row (stable: file order inside a row); already grouped when no id repeats out of order
        f->row_seg.assign(R + 1, 0);
        for (const Seg &s : raw) f->row_seg[(size_t)s.row + 1]++;
        for (size_t r = 0; r < R; r++) f->row_seg[r + 1] += f->row_seg[r];
        bool sorted = true;
        for (size_t i = 1; i < raw.size(); i++) if (raw[i].row < raw[i - 1].row) { sorted = false; break; }
        if (sorted) f->segs.swap(raw);
        else {
            f->segs.resize(raw.size());
            std::vector<int64_t> at(f->row_seg.begin(), f->row_seg.end() - 1);
            for (const Seg &s : raw) f->segs[(size_t)at[(size_t)s.row]++] = s;
        }
        f->row_off.assign(R + 1, 0);
        f->id_bytes = 0;
        for (size_t r = 0; r < R; r++) {
            int64_t len = 0;
            for (int64_t i = f->row_seg[r]; i < f->row_seg[r + 1]; i++) len += f->segs[(size_t)i].len;
            f->row_off[r + 1] = f->row_off[r] + len;
            f->id_bytes += f->id_len[r];
        }
    }
    tr.lap("group");
    if (R > 0x7fffffffULL - 1) return ffail(f, MP_ERR_ARG, "too many records");
    return MP_OK;
}

}  // namespace

mp_fasta::~mp_fasta() {
    if (own) munmap(own, own_bytes);
}

extern "C" {

const char *mp_fasta_error(const mp_fasta *f) { return f ? f->err : "mp_fasta_parse failed"; }
void mp_fasta_destroy(mp_fasta *f) { delete f; }

int mp_fasta_parse_buffer(const uint8_t *bytes, int64_t n_bytes, int32_t n_threads, mp_fasta **out) {
    if (!out) return MP_ERR_ARG;
    *out = nullptr;
    if (n_bytes < 0 || (n_bytes && !bytes)) return MP_ERR_ARG;
    mp_fasta *f = new (std::nothrow) mp_fasta();
    if (!f) return MP_ERR_NOMEM;
    *out = f;
    f->own = big_alloc((size_t)n_bytes + 1, &f->own_bytes, false);
    if (!f->own) return ffail(f, MP_ERR_NOMEM, "out of memory (%lld bytes)", (long long)n_bytes);
    if (n_bytes) memcpy(f->own, bytes, (size_t)n_bytes);
    f->buf = f->own;
    f->n = n_bytes;
    f->n_threads = threads_for(n_threads, n_bytes);
    return parse(f);
}

int mp_fasta_parse_file(const char *path, int32_t n_threads, mp_fasta **out) {
    if (!out) return MP_ERR_ARG;
    *out = nullptr;
    if (!path) return MP_ERR_ARG;
    mp_fasta *f = new (std::nothrow) mp_fasta();
    if (!f) return MP_ERR_NOMEM;
    *out = f;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return ffail(f, MP_ERR_ARG, "%s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return ffail(f, MP_ERR_ARG, "%s: %s", path, strerror(errno)); }
    int64_t n = (int64_t)st.st_size;
    std::vector<uint8_t> piped;
    if (!S_ISREG(st.st_mode)) {
        // a pipe / device: read to the end
        uint8_t tmp[1 << 16];
        ssize_t got;
        while ((got = read(fd, tmp, sizeof tmp)) > 0) piped.insert(piped.end(), tmp, tmp + got);
        n = (int64_t)piped.size();
    }
    // A regular file is mapped read-only: no copy and no fresh pages to fault in — the scan threads are the first to touch the
    // page cache (the file must not shrink while the handle lives).  MP_HOST_READ=pread reads it into private memory instead,
    // =huge into huge-page memory.
    const char *mode = getenv("MP_HOST_READ");
    const bool want_map = (!mode || !strcmp(mode, "mmap")) && S_ISREG(st.st_mode) && n > 0;
    if (want_map) {
        void *m = mmap(nullptr, (size_t)n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            f->own = (uint8_t *)m;
            f->own_bytes = (size_t)n;
            f->n_threads = threads_for(n_threads, n);
            f->buf = f->own;
            f->n = n;
            return parse(f);
        }
    }
    f->own = big_alloc((size_t)n + 1, &f->own_bytes, mode && !strcmp(mode, "huge"));
    if (!f->own) { close(fd); return ffail(f, MP_ERR_NOMEM, "out of memory (%lld bytes)", (long long)n); }
    f->n_threads = threads_for(n_threads, n);
    Trace tr;
    if (!piped.empty()) memcpy(f->own, piped.data(), (size_t)n);
    else if (S_ISREG(st.st_mode) && n) {
        const int T = f->n_threads;
        std::vector<int> bad((size_t)T, 0);
        auto rd = [&](int t) {
            int64_t a = n * t / T, z = n * (t + 1) / T;
            while (a < z) {
                ssize_t got = pread(fd, f->own + a, (size_t)(z - a), (off_t)a);
                if (got <= 0) { bad[(size_t)t] = 1; return; }
                a += got;
            }
        };
        if (T == 1) rd(0);
        else {
            mp::run_on_threads(T, rd);
        }
        for (int x : bad) if (x) { close(fd); return ffail(f, MP_ERR_ARG, "%s: short read", path); }
    }
    close(fd);
    tr.lap("read");
    f->buf = f->own;
    f->n = n;
    return parse(f);
}

int mp_file_count_newlines(const char *path, int32_t n_threads, int64_t *count) {
    if (!path || !count) return MP_ERR_ARG;
    *count = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return MP_ERR_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return MP_ERR_ARG; }
    // newlines of Python's text mode: \n, \r\n (one) and a lone \r
    auto tally = [](const uint8_t *b, int64_t m, bool more, uint8_t next) {      // [0, m) counted; b[m] = next when `more`
        int64_t c = std::count(b, b + m, (uint8_t)'\n');
        const uint8_t *p = b, *e = b + m;
        while ((p = (const uint8_t *)memchr(p, '\r', (size_t)(e - p)))) {
            const bool last = p + 1 == e;
            if (last ? !(more && next == '\n') : p[1] != '\n') c++;
            p++;
        }
        return c;
    };
    if (!S_ISREG(st.st_mode)) {                                  // a pipe / device: read to the end, one pass
        std::vector<uint8_t> all;
        uint8_t tmp[1 << 16];
        ssize_t got;
        while ((got = read(fd, tmp, sizeof tmp)) > 0) all.insert(all.end(), tmp, tmp + got);
        close(fd);
        *count = tally(all.data(), (int64_t)all.size(), false, 0);
        return MP_OK;
    }
    const int64_t n = (int64_t)st.st_size;
    const int T = threads_for(n_threads, n);
    std::vector<int64_t> part((size_t)T, 0);
    std::vector<int> bad((size_t)T, 0);
    auto run = [&](int t) {
        std::vector<uint8_t> tmp((4 << 20) + 1);
        int64_t a = n * t / T;
        const int64_t z = n * (t + 1) / T;
        while (a < z) {
            const int64_t m = std::min<int64_t>(4 << 20, z - a), want = std::min<int64_t>(m + 1, n - a);   // one byte of lookahead
            int64_t have = 0;
            while (have < want) {
                ssize_t got = pread(fd, tmp.data() + have, (size_t)(want - have), (off_t)(a + have));
                if (got <= 0) { bad[(size_t)t] = 1; return; }
                have += got;
            }
            part[(size_t)t] += tally(tmp.data(), m, want > m, want > m ? tmp[(size_t)m] : 0);
            a += m;
        }
    };
    if (T == 1) run(0);
    else {
        mp::run_on_threads(T, run);
    }
    close(fd);
    for (int x : bad) if (x) return MP_ERR_ARG;
    for (int64_t c : part) *count += c;
    return MP_OK;
}

int mp_fasta_sizes(const mp_fasta *f, int32_t *n_rows, int64_t *n_residue_bytes, int64_t *n_id_bytes) {
    if (!f) return MP_ERR_ARG;
    if (n_rows) *n_rows = (int32_t)f->id_off_src.size();
    if (n_residue_bytes) *n_residue_bytes = f->row_off.empty() ? 0 : f->row_off.back();
    if (n_id_bytes) *n_id_bytes = f->id_bytes;
    return MP_OK;
}

int mp_fasta_rows(const mp_fasta *f, uint8_t *data, int64_t *row_off) {
    if (!f) return MP_ERR_ARG;
    const size_t R = f->id_off_src.size();
    if (row_off) memcpy(row_off, f->row_off.data(), sizeof(int64_t) * (R + 1));
    if (!data || R == 0) return MP_OK;
    const int T = std::max(1, std::min<int>(f->n_threads, (int)(f->row_off.back() / (1 << 20) + 1)));
    auto copy = [&](int t) {
        size_t r0 = R * (size_t)t / (size_t)T, r1 = R * ((size_t)t + 1) / (size_t)T;
        for (size_t r = r0; r < r1; r++) {
            uint8_t *dst = data + f->row_off[r];
            for (int64_t i = f->row_seg[r]; i < f->row_seg[r + 1]; i++) {
                const Seg &s = f->segs[(size_t)i];
                memcpy(dst, f->buf + s.off, (size_t)s.len);
                dst += s.len;
            }
        }
    };
    if (T == 1) copy(0);
    else {
        mp::run_on_threads(T, copy);
    }
    return MP_OK;
}

// residue bytes [byte0, byte1) of the rows laid end to end (what mp_fasta_rows writes at data[byte0 .. byte1)), on n_threads threads:
// the streamed load (mp_load_msa_fasta, pack.hip) fills its transfer buffers with it chunk by chunk
int mp_fasta_gather(const mp_fasta *f, int64_t byte0, int64_t byte1, uint8_t *dst, int32_t n_threads) {
    if (!f || !dst) return MP_ERR_ARG;
    const size_t R = f->id_off_src.size();
    if (R == 0 || byte1 <= byte0) return MP_OK;
    if (byte0 < 0 || byte1 > f->row_off[R]) return MP_ERR_ARG;
    // rows that overlap the range
    const size_t ra = (size_t)(std::upper_bound(f->row_off.begin(), f->row_off.end(), byte0) - f->row_off.begin()) - 1;
    const size_t rb = (size_t)(std::lower_bound(f->row_off.begin(), f->row_off.end(), byte1) - f->row_off.begin());      // rows [ra, rb)
    const int T = std::max(1, std::min<int>(n_threads > 0 ? n_threads : f->n_threads, (int)((byte1 - byte0) / (1 << 20) + 1)));
    auto copy = [&](int t) {
        const size_t r0 = ra + (rb - ra) * (size_t)t / (size_t)T, r1 = ra + (rb - ra) * ((size_t)t + 1) / (size_t)T;
        for (size_t r = r0; r < r1; r++) {
            int64_t at = f->row_off[r];                            // position of the segment's first byte in the concatenation
            for (int64_t i = f->row_seg[r]; i < f->row_seg[r + 1]; i++) {
                const Seg &s = f->segs[(size_t)i];
                const int64_t a = std::max(at, byte0), b = std::min<int64_t>(at + s.len, byte1);
                if (b > a) memcpy(dst + (a - byte0), f->buf + s.off + (a - at), (size_t)(b - a));
                at += s.len;
            }
        }
    };
    if (T == 1) copy(0);
    else mp::run_on_threads(T, copy);
    return MP_OK;
}

int mp_fasta_ids(const mp_fasta *f, uint8_t *ids, int64_t *id_off) {
    if (!f || !id_off) return MP_ERR_ARG;
    const size_t R = f->id_off_src.size();
    int64_t o = 0;
    for (size_t r = 0; r < R; r++) {
        id_off[r] = o;
        if (ids) memcpy(ids + o, f->buf + f->id_off_src[r], (size_t)f->id_len[r]);
        o += f->id_len[r];
    }
    id_off[R] = o;
    return MP_OK;
}

}  // extern "C"
