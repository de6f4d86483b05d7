// Host-side netlist tokenizer for large, regular CSV files.
//
// One pass over the file: splits lines and fields the way the reference's
// csv.reader(skipinitialspace=True) does for UNQUOTED input, checks every row against the
// reference's check_input rules (known type, exact field count, numeric value), numbers the
// node labels in order of first appearance (anode before bnode -- the dict insertion order
// the reference's node numbering depends on, reference nodal/nodal.py:222-257) and rejects
// duplicated component names.  Anything else -- quotes, macros (OPMODEL / OPAMP), malformed
// rows, exotic number spellings -- returns a non-zero status and the Python side falls back
// to the exact row-by-row parser, which then raises the reference's own exception.
//
// This is front-end plumbing (strings -> integers), not part of the GPU hot path; it is
// built with g++ into nodal_amd/libnodal_csv.so and is optional (fastparse.py falls back to
// the pandas reader without it).
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string_view>
#include <thread>
#include <utility>
#include <vector>

extern "C" {

typedef struct {
    int64_t nrows, nnodes;
    int32_t status;     // 0 = ok, otherwise the reason the file is "irregular"
    int64_t bad_line;   // 0-based physical line of the first irregularity
    int64_t *line_off;  // per component row: offset / length of its line in the buffer
    int32_t *line_len;
    uint8_t *type_idx;  // index into {"R","A","E","VCVS","VCCS","CCVS","CCCS"}
    uint8_t *nfields;
    double *value;
    int32_t *acode, *bcode;  // node ids in first-appearance order
    char *names_blob;        // component names joined by '\n'
    int64_t names_bytes;
    char *labels_blob;       // node labels joined by '\n', in id order
    int64_t labels_bytes;
} nodal_csv_result;

enum {
    CSV_OK = 0,
    CSV_QUOTES = 1,
    CSV_BLANK_WITH_SPACES = 2,
    CSV_EMPTY_FIRST_FIELD = 3,
    CSV_TOO_MANY_FIELDS = 4,
    CSV_UNKNOWN_TYPE = 5,
    CSV_FIELD_COUNT = 6,
    CSV_BAD_VALUE = 7,
    CSV_DUPLICATE_NAME = 8,
    CSV_NO_COMPONENTS = 9,
    CSV_NEWLINE_IN_FIELD = 10,
    CSV_NO_MEMORY = 11,
};

void nodal_csv_free(nodal_csv_result *r) {
    if (!r) return;
    free(r->line_off); free(r->line_len); free(r->type_idx); free(r->nfields); free(r->value);
    free(r->acode); free(r->bcode); free(r->names_blob); free(r->labels_blob);
    memset(r, 0, sizeof *r);
}

}  // extern "C"

static const char *const TYPE_NAMES[7] = {"R", "A", "E", "VCVS", "VCCS", "CCVS", "CCCS"};
static const int TYPE_FIELDS[7] = {5, 5, 5, 7, 7, 8, 8};

static int type_index(std::string_view s) {
    for (int i = 0; i < 7; ++i)
        if (s == TYPE_NAMES[i]) return i;
    return -1;
}

// Open-addressing string -> dense id table (string_views into the file buffer): the node-based
// std::unordered containers cost 2 s on the 2e6-row grid(1000) netlist, this one 0.3 s.
struct StringIds {
    std::vector<int32_t> slots;  // id + 1, 0 = empty
    std::vector<uint64_t> hashes;
    std::vector<std::string_view> items;
    uint64_t mask = 0;
    explicit StringIds(size_t expected) {
        size_t cap = 64;
        while (cap < 2 * expected) cap <<= 1;
        slots.assign(cap, 0);
        mask = cap - 1;
        hashes.reserve(expected);
        items.reserve(expected);
    }
    static uint64_t hash(std::string_view s) {
        const char *p = s.data();
        size_t n = s.size();
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        while (n >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdull;
            h ^= h >> 32;
            p += 8;
            n -= 8;
        }
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
        return h ^ (h >> 29);
    }
    void grow() {
        const size_t cap = slots.size() * 2;
        slots.assign(cap, 0);
        mask = cap - 1;
        for (size_t e = 0; e < items.size(); ++e) {
            uint64_t i = hashes[e] & mask;
            while (slots[i]) i = (i + 1) & mask;
            slots[i] = (int32_t)e + 1;
        }
    }
    // (the table of a 2e6-row file is 50 MB: every probe of a new name is a cache miss unless the slot
    // was requested a few rows ahead -- see the two-phase loop of nodal_csv_parse)
    void prefetch(uint64_t h) const { __builtin_prefetch(&slots[h & mask]); }
    // id of s (hash h), inserting it if new (*inserted tells which)
    int32_t get(std::string_view s, uint64_t h, bool *inserted) {
        uint64_t i = h & mask;
        while (slots[i]) {
            const int32_t e = slots[i] - 1;
            if (hashes[e] == h && items[e] == s) {
                *inserted = false;
                return e;
            }
            i = (i + 1) & mask;
        }
        if (2 * (items.size() + 1) > slots.size()) {
            grow();
            i = h & mask;
            while (slots[i]) i = (i + 1) & mask;
        }
        const int32_t e = (int32_t)items.size();
        slots[i] = e + 1;
        hashes.push_back(h);
        items.push_back(s);
        *inserted = true;
        return e;
    }
};

// [+-]digits[.digits] with at most 15 significant digits: mantissa and power of ten are both
// exact doubles, so one IEEE division gives the correctly rounded value (what float() returns).
static bool parse_plain_decimal(std::string_view v, double *out) {
    static const double P10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    size_t i = 0;
    bool neg = false;
    if (i < v.size() && (v[i] == '+' || v[i] == '-')) neg = v[i++] == '-';
    uint64_t m = 0;
    int digits = 0, frac = 0;
    bool dot = false, any = false;
    for (; i < v.size(); ++i) {
        const char ch = v[i];
        if (ch >= '0' && ch <= '9') {
            any = true;
            if (digits == 15) return false;
            if (m != 0 || ch != '0') ++digits;
            m = m * 10 + (uint64_t)(ch - '0');
            if (dot) ++frac;
        } else if (ch == '.' && !dot) dot = true;
        else return false;
    }
    if (!any || frac > 15) return false;
    const double r = (double)m / P10[frac];
    *out = neg ? -r : r;
    return true;
}

// ---- one chunk of the file (whole lines), tokenized on its own thread --------------------------------------------
// Round 5: the file is cut at newlines into one chunk per host thread.  A chunk splits its lines, checks them and
// numbers ITS node labels in order of first appearance in a table of its own; afterwards the chunks' label lists are
// merged in file order -- a label is new to the file in the first chunk that holds it, and inside a chunk the new
// labels keep their local order, which is exactly the order of first appearance in the file (anode before bnode,
// rows in file order: the dict insertion order the reference's numbering depends on, nodal/nodal.py:222-257).
// Both merges -- labels, and the uniqueness of the component names -- run on the same threads, split by hash bucket.
struct Chunk {
    int64_t begin = 0, end = 0;   // byte range [begin, end): whole lines
    int64_t nlines = 0;           // physical lines read (all of them unless the chunk stopped at an irregular line)
    int status = CSV_OK;          // the chunk's first irregular line, if any (it stops there)
    int64_t bad_local = 0;
    std::vector<int64_t> line_off, row_line;  // per component row: offset of its line; local physical line number
    std::vector<int32_t> line_len, la, lb;    // ...: length of its line; LOCAL ids of its two lead labels
    std::vector<uint8_t> type_idx, nfields;
    std::vector<double> value;
    std::vector<std::string_view> name;
    std::vector<uint64_t> hname;
    StringIds labels{64};
    // filled by the merge
    std::vector<int32_t> gid;                 // local label id -> id in the file's first-appearance order
    std::vector<uint8_t> is_new;              // the label appears in no earlier chunk
    std::vector<std::pair<int32_t, int32_t>> canon;  // (chunk, local id) of its first appearance otherwise
    int64_t line_base = 0, row_base = 0, new_base = 0, new_count = 0, name_bytes = 0, label_bytes = 0;
};

static void tokenize_chunk(const char *buf, Chunk &ck) {
    const int64_t len = ck.end;
    const size_t guess = (size_t)((ck.end - ck.begin) / 24 + 16);
    ck.line_off.reserve(guess); ck.row_line.reserve(guess); ck.line_len.reserve(guess); ck.la.reserve(guess);
    ck.lb.reserve(guess); ck.type_idx.reserve(guess); ck.nfields.reserve(guess); ck.value.reserve(guess);
    ck.name.reserve(guess); ck.hname.reserve(guess);
    ck.labels = StringIds(guess / 2);
    auto fail = [&](int status, int64_t line) {
        ck.status = status;
        ck.bad_local = line;
    };
    int64_t pos = ck.begin, lineno = 0;
    char numbuf[64];
    while (pos < len) {
        const char *nl = static_cast<const char *>(memchr(buf + pos, '\n', (size_t)(len - pos)));
        int64_t end = nl ? nl - buf : len;
        const int64_t next = nl ? end + 1 : len;
        const int64_t start = pos;
        pos = next;
        const int64_t this_line = lineno++;
        ck.nlines = lineno;
        if (end > start && buf[end - 1] == '\r') --end;
        if (end == start) continue;  // empty line: csv.reader yields [] and the reference skips it
        bool blank = true;
        for (int64_t i = start; i < end; ++i) {
            const char ch = buf[i];
            if (ch == '"') return fail(CSV_QUOTES, this_line);
            if (ch == '\r') return fail(CSV_NEWLINE_IN_FIELD, this_line);
            if (ch != ' ' && ch != '\t') blank = false;
        }
        if (blank) return fail(CSV_BLANK_WITH_SPACES, this_line);  // the reference raises IndexError there
        // split (skipinitialspace: blanks right after a delimiter / at the start are dropped)
        std::string_view f[9];
        int nf = 0;
        int64_t i = start;
        while (true) {
            while (i < end && buf[i] == ' ') ++i;
            int64_t j = i;
            while (j < end && buf[j] != ',') ++j;
            if (nf == 9) break;
            f[nf++] = std::string_view(buf + i, (size_t)(j - i));
            if (j >= end) break;
            i = j + 1;
            if (i == end) {  // trailing comma: one more, empty, field
                if (nf < 9) f[nf++] = std::string_view(buf + end, 0);
                break;
            }
        }
        if (f[0].empty()) return fail(CSV_EMPTY_FIRST_FIELD, this_line);
        if (f[0][0] == '#') continue;  // comment row
        if (nf > 8) return fail(CSV_TOO_MANY_FIELDS, this_line);
        if (nf < 2) return fail(CSV_FIELD_COUNT, this_line);
        const int ti = type_index(f[1]);
        if (ti < 0) return fail(CSV_UNKNOWN_TYPE, this_line);
        if (nf != TYPE_FIELDS[ti]) return fail(CSV_FIELD_COUNT, this_line);
        // value: plain decimal spellings only; everything float() accepts beyond that
        // ("1_0", " 1 ", "nan", "inf") goes to the exact parser
        const std::string_view v = f[2];
        if (v.empty() || v.size() >= sizeof numbuf) return fail(CSV_BAD_VALUE, this_line);
        bool digit = false;
        for (char ch : v) {
            if (ch >= '0' && ch <= '9') digit = true;
            else if (ch != '+' && ch != '-' && ch != '.' && ch != 'e' && ch != 'E')
                return fail(CSV_BAD_VALUE, this_line);
        }
        if (!digit) return fail(CSV_BAD_VALUE, this_line);
        double val;
        if (!parse_plain_decimal(v, &val)) {  // exponent forms, long mantissas: strtod (correctly rounded too)
            memcpy(numbuf, v.data(), v.size());
            numbuf[v.size()] = 0;
            char *endp = nullptr;
            errno = 0;
            val = strtod(numbuf, &endp);
            if (endp != numbuf + v.size()) return fail(CSV_BAD_VALUE, this_line);
        }
        bool fresh = false;
        ck.la.push_back(ck.labels.get(f[3], StringIds::hash(f[3]), &fresh));  // anode before bnode
        ck.lb.push_back(ck.labels.get(f[4], StringIds::hash(f[4]), &fresh));
        ck.name.push_back(f[0]);
        ck.hname.push_back(StringIds::hash(f[0]));
        ck.row_line.push_back(this_line);
        ck.line_off.push_back(start);
        ck.line_len.push_back((int32_t)(end - start));
        ck.type_idx.push_back((uint8_t)ti);
        ck.nfields.push_back((uint8_t)nf);
        ck.value.push_back(val);
    }
}

// f(k) for k in [0, count) on up to `threads` host threads (the calling one included)
template <class F>
static void parallel_for(int count, int threads, F f) {
    if (threads <= 1 || count <= 1) {
        for (int k = 0; k < count; ++k) f(k);
        return;
    }
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int k = next.fetch_add(1); k < count; k = next.fetch_add(1)) f(k);
    };
    std::vector<std::thread> th;
    const int extra = (threads < count ? threads : count) - 1;
    for (int t = 0; t < extra; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

extern "C" {

int nodal_csv_parse(const char *buf, int64_t len, nodal_csv_result *out) {
    memset(out, 0, sizeof *out);
    auto fail = [&](int status, int64_t line) {
        out->status = status;
        out->bad_line = line;
        return status;
    };
    // ---- chunks ----
    int threads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("NODAL_HOST_THREADS")) threads = atoi(e);
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    // (four chunks per thread: the threads stay level, and a chunk's label table stays in the L2 cache)
    int nchunks = threads > 1 ? 4 * threads : 1;
    if (const char *e = getenv("NODAL_CSV_CHUNKS")) nchunks = atoi(e);  // (testing: more chunks than a small file deserves)
    else if (len < (int64_t)nchunks * (1 << 18)) nchunks = (int)(len >> 18);
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 256) nchunks = 256;
    std::vector<Chunk> chunks((size_t)nchunks);
    {
        int64_t at = 0;
        for (int c = 0; c < nchunks; ++c) {
            chunks[(size_t)c].begin = at;
            int64_t stop = c + 1 == nchunks ? len : len * (c + 1) / nchunks;
            if (stop < at) stop = at;
            if (c + 1 < nchunks && stop < len) {  // to the end of the line the cut falls into
                const char *nl = static_cast<const char *>(memchr(buf + stop, '\n', (size_t)(len - stop)));
                stop = nl ? (nl - buf) + 1 : len;
            }
            chunks[(size_t)c].end = stop;
            at = stop;
        }
    }
    parallel_for(nchunks, threads, [&](int c) { tokenize_chunk(buf, chunks[(size_t)c]); });
    // physical line numbers, row numbers; the first irregular line among the chunks (later ones do not count)
    int first_bad = -1;
    {
        int64_t lines = 0, rows = 0;
        for (int c = 0; c < nchunks; ++c) {
            Chunk &ck = chunks[(size_t)c];
            ck.line_base = lines;
            ck.row_base = rows;
            lines += ck.nlines;
            rows += (int64_t)ck.name.size();
            if (ck.status != CSV_OK) {
                first_bad = c;
                break;
            }
        }
    }
    const int live = first_bad >= 0 ? first_bad + 1 : nchunks;  // chunks whose rows count
    // ---- component names must be unique: the rows by hash bucket, every bucket in file order on one thread ----
    constexpr int BUCKETS = 32;
    auto bucket_of = [](uint64_t h) { return (int)(h >> 59); };  // (the table slots use the low bits)
    int64_t dup_line[BUCKETS];
    {
        int64_t total_rows = 0;
        for (int c = 0; c < live; ++c) total_rows += (int64_t)chunks[(size_t)c].name.size();
        parallel_for(BUCKETS, threads, [&](int b) {
            dup_line[b] = -1;
            StringIds seen((size_t)(total_rows / BUCKETS + total_rows / (4 * BUCKETS) + 16));
            for (int c = 0; c < live && dup_line[b] < 0; ++c) {
                const Chunk &ck = chunks[(size_t)c];
                const size_t nrow = ck.name.size();
                for (size_t r = 0; r < nrow; ++r) {
                    if (bucket_of(ck.hname[r]) != b) continue;
                    bool fresh = false;
                    seen.get(ck.name[r], ck.hname[r], &fresh);
                    if (!fresh) {
                        dup_line[b] = ck.line_base + ck.row_line[r];
                        break;
                    }
                }
            }
        });
    }
    {
        // the FIRST irregular line of the file: the smallest of the chunks' own and the first repeated name
        int64_t best_line = -1;
        int best_status = CSV_OK;
        if (first_bad >= 0) {
            best_line = chunks[(size_t)first_bad].line_base + chunks[(size_t)first_bad].bad_local;
            best_status = chunks[(size_t)first_bad].status;
        }
        for (int b = 0; b < BUCKETS; ++b)
            if (dup_line[b] >= 0 && (best_line < 0 || dup_line[b] < best_line)) {
                best_line = dup_line[b];
                best_status = CSV_DUPLICATE_NAME;
            }
        if (best_status != CSV_OK) return fail(best_status, best_line);
    }
    int64_t nrows = 0;
    for (const Chunk &ck : chunks) nrows += (int64_t)ck.name.size();
    if (nrows == 0) return fail(CSV_NO_COMPONENTS, 0);
    // ---- node labels: which chunk saw each one first ----
    for (Chunk &ck : chunks) {
        ck.is_new.assign(ck.labels.items.size(), 0);
        ck.canon.assign(ck.labels.items.size(), {-1, -1});
        ck.gid.assign(ck.labels.items.size(), -1);
    }
    {
        int64_t total_labels = 0;
        for (const Chunk &ck : chunks) total_labels += (int64_t)ck.labels.items.size();
        parallel_for(BUCKETS, threads, [&](int b) {
            // first[label] = (chunk, local id) of its first appearance, for the labels of this bucket
            StringIds seen((size_t)(total_labels / BUCKETS + total_labels / (4 * BUCKETS) + 16));
            std::vector<std::pair<int32_t, int32_t>> first;
            for (int c = 0; c < nchunks; ++c) {
                Chunk &ck = chunks[(size_t)c];
                const size_t m = ck.labels.items.size();
                for (size_t l = 0; l < m; ++l) {
                    const uint64_t h = ck.labels.hashes[l];
                    if (bucket_of(h) != b) continue;
                    bool fresh = false;
                    const int32_t e = seen.get(ck.labels.items[l], h, &fresh);
                    if (fresh) {
                        first.push_back({(int32_t)c, (int32_t)l});
                        ck.is_new[l] = 1;
                    } else {
                        ck.canon[l] = first[(size_t)e];
                    }
                }
            }
        });
    }
    // ids in first-appearance order: chunk by chunk, the new labels of a chunk in its local order
    {
        int64_t base = 0;
        for (Chunk &ck : chunks) {
            ck.new_base = base;
            int64_t k = 0, bytes = 0;
            for (size_t l = 0; l < ck.is_new.size(); ++l)
                if (ck.is_new[l]) {
                    ck.gid[l] = (int32_t)(base + k++);
                    bytes += (int64_t)ck.labels.items[l].size() + 1;
                }
            ck.new_count = k;
            ck.label_bytes = bytes;
            base += k;
        }
        out->nnodes = base;
    }
    parallel_for(nchunks, threads, [&](int c) {
        Chunk &ck = chunks[(size_t)c];
        for (size_t l = 0; l < ck.gid.size(); ++l)
            if (!ck.is_new[l]) ck.gid[l] = chunks[(size_t)ck.canon[l].first].gid[(size_t)ck.canon[l].second];
        int64_t bytes = 0;
        for (const auto &nm : ck.name) bytes += (int64_t)nm.size() + 1;
        ck.name_bytes = bytes;
    });
    // ---- output arrays ----
    int64_t names_total = 0, labels_total = 0;
    std::vector<int64_t> name_at((size_t)nchunks), label_at((size_t)nchunks);
    for (int c = 0; c < nchunks; ++c) {
        name_at[(size_t)c] = names_total;
        label_at[(size_t)c] = labels_total;
        names_total += chunks[(size_t)c].name_bytes;
        labels_total += chunks[(size_t)c].label_bytes;
    }
    out->nrows = nrows;
    out->line_off = static_cast<int64_t *>(malloc((size_t)nrows * 8 + 8));
    out->line_len = static_cast<int32_t *>(malloc((size_t)nrows * 4 + 8));
    out->type_idx = static_cast<uint8_t *>(malloc((size_t)nrows + 8));
    out->nfields = static_cast<uint8_t *>(malloc((size_t)nrows + 8));
    out->value = static_cast<double *>(malloc((size_t)nrows * 8 + 8));
    out->acode = static_cast<int32_t *>(malloc((size_t)nrows * 4 + 8));
    out->bcode = static_cast<int32_t *>(malloc((size_t)nrows * 4 + 8));
    out->names_blob = static_cast<char *>(malloc((size_t)names_total + 8));
    out->labels_blob = static_cast<char *>(malloc((size_t)labels_total + 8));
    if (!out->line_off || !out->line_len || !out->type_idx || !out->nfields || !out->value || !out->acode ||
        !out->bcode || !out->names_blob || !out->labels_blob) {
        nodal_csv_free(out);
        out->status = CSV_NO_MEMORY;
        return CSV_NO_MEMORY;
    }
    parallel_for(nchunks, threads, [&](int c) {
        const Chunk &ck = chunks[(size_t)c];
        const size_t nrow = ck.name.size();
        const int64_t r0 = ck.row_base;
        if (nrow) {
            memcpy(out->line_off + r0, ck.line_off.data(), nrow * 8);
            memcpy(out->line_len + r0, ck.line_len.data(), nrow * 4);
            memcpy(out->type_idx + r0, ck.type_idx.data(), nrow);
            memcpy(out->nfields + r0, ck.nfields.data(), nrow);
            memcpy(out->value + r0, ck.value.data(), nrow * 8);
        }
        char *np = out->names_blob + name_at[(size_t)c];
        for (size_t r = 0; r < nrow; ++r) {
            out->acode[r0 + (int64_t)r] = ck.gid[(size_t)ck.la[r]];
            out->bcode[r0 + (int64_t)r] = ck.gid[(size_t)ck.lb[r]];
            memcpy(np, ck.name[r].data(), ck.name[r].size());
            np += ck.name[r].size();
            *np++ = '\n';
        }
        char *lp = out->labels_blob + label_at[(size_t)c];
        for (size_t l = 0; l < ck.is_new.size(); ++l)
            if (ck.is_new[l]) {
                const std::string_view &s = ck.labels.items[l];
                memcpy(lp, s.data(), s.size());
                lp += s.size();
                *lp++ = '\n';
            }
    });
    out->names_bytes = names_total ? names_total - 1 : 0;   // without the last separator
    out->labels_bytes = labels_total ? labels_total - 1 : 0;
    return CSV_OK;
}

}  // extern "C"

// ---- Solution.__str__ at 1e6 nodes (reference nodal/nodal.py:422-434) ---------------------
// The reference prints "e(name) \t= value" for the node names in sorted() order -- lexicographic STRING order, "10" < "2"
// -- with str(np.float64), i.e. the shortest digits that round-trip, laid out by Python's repr rule.  At 1e6 nodes that
// is a sort of a million strings and a million float formats: 0.55 s in Python, the largest item of the user-visible
// path once the front end took 0.08 s.  Here: the labels (the tokenizer's blob, ids in first-appearance order) are
// bucketed by their first two bytes, the buckets sorted on host threads (byte order = code-point order for UTF-8, which
// is what Python's str comparison is), the lines formatted on threads into one buffer.
#include <algorithm>
#include <charconv>
#include <cmath>

// repr(float) of Python 3 (float_repr_style 'short'): shortest round-trip digits; exponent form iff decpt <= -4 or
// decpt > 16 (decpt: position of the decimal point relative to the digit string), "e-05" style exponents, ".0" appended
// to integral values in fixed form.  Returns the number of characters written (buf holds at least 32).
static int python_repr(double v, char *buf) {
    if (std::isnan(v)) { memcpy(buf, "nan", 3); return 3; }
    if (std::isinf(v)) { const char *s = v < 0 ? "-inf" : "inf"; const int n = v < 0 ? 4 : 3; memcpy(buf, s, (size_t)n); return n; }
    char sci[40];
    auto res = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX, shortest
    const int len = (int)(res.ptr - sci);
    int p = 0, o = 0;
    if (sci[0] == '-') { buf[o++] = '-'; p = 1; }
    char digits[24];
    int nd = 0;
    for (; p < len && sci[p] != 'e'; ++p)
        if (sci[p] != '.') digits[nd++] = sci[p];
    int e10 = 0;
    {
        ++p;  // 'e'
        const bool neg = sci[p] == '-';
        ++p;
        for (; p < len; ++p) e10 = e10 * 10 + (sci[p] - '0');
        if (neg) e10 = -e10;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;  // (to_chars gives no trailing zeros; belt and braces)
    const int decpt = e10 + 1;
    if (decpt <= -4 || decpt > 16) {
        buf[o++] = digits[0];
        if (nd > 1) {
            buf[o++] = '.';
            memcpy(buf + o, digits + 1, (size_t)(nd - 1));
            o += nd - 1;
        }
        buf[o++] = 'e';
        int e = decpt - 1;
        buf[o++] = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) buf[o++] = eb[--ne];
        return o;
    }
    if (decpt <= 0) {
        buf[o++] = '0';
        buf[o++] = '.';
        for (int z = 0; z < -decpt; ++z) buf[o++] = '0';
        memcpy(buf + o, digits, (size_t)nd);
        return o + nd;
    }
    if (decpt >= nd) {
        memcpy(buf + o, digits, (size_t)nd);
        o += nd;
        for (int z = 0; z < decpt - nd; ++z) buf[o++] = '0';
        buf[o++] = '.';
        buf[o++] = '0';
        return o;
    }
    memcpy(buf + o, digits, (size_t)decpt);
    o += decpt;
    buf[o++] = '.';
    memcpy(buf + o, digits + decpt, (size_t)(nd - decpt));
    return o + (nd - decpt);
}

extern "C" {

// repr of one double (testing hook: the formatter against Python's own repr)
int nodal_repr_double(double v, char *buf32) { return python_repr(v, buf32); }

// "prefix" + name + ") \t= " + repr(values[index_of[k]]) for every label k with index_of[k] >= 0, the lines in sorted()
// order of the names, joined by '\n' (no trailing newline).  labels: nlabels names joined by '\n'.  *out is malloc'ed
// (nodal_csv_free_buffer).  Returns 0, or CSV_NO_MEMORY.
int nodal_format_lines(const char *labels, int64_t labels_len, int64_t nlabels, const int64_t *index_of,
                       const double *values, const char *prefix, char **out, int64_t *out_len) {
    *out = nullptr;
    *out_len = 0;
    int threads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("NODAL_HOST_THREADS")) threads = atoi(e);
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (nlabels < 20000) threads = 1;
    const size_t plen = strlen(prefix);
    // label k = bytes [off[k], off[k + 1] - 1)
    std::vector<int64_t> off((size_t)nlabels + 1);
    {
        int64_t k = 0, at = 0;
        off[0] = 0;
        while (k < nlabels) {
            const char *nl = static_cast<const char *>(memchr(labels + at, '\n', (size_t)(labels_len - at)));
            const int64_t end = nl ? nl - labels : labels_len;
            off[(size_t)++k] = end + 1;
            at = end + 1;
            if (!nl) break;
        }
        if (k != nlabels) return CSV_FIELD_COUNT;
    }
    auto name_of = [&](int64_t k) { return std::string_view(labels + off[(size_t)k], (size_t)(off[(size_t)k + 1] - 1 - off[(size_t)k])); };
    // buckets by the first two bytes (0 for a missing byte: a shorter name sorts first)
    auto key_of = [&](int64_t k) {
        const std::string_view s = name_of(k);
        const unsigned b0 = s.size() > 0 ? (unsigned char)s[0] : 0u, b1 = s.size() > 1 ? (unsigned char)s[1] : 0u;
        return (b0 << 8) | b1;
    };
    std::vector<int64_t> start(65537, 0);
    std::vector<int32_t> order;
    int64_t kept = 0;
    for (int64_t k = 0; k < nlabels; ++k)
        if (index_of[k] >= 0) {
            ++start[(size_t)key_of(k) + 1];
            ++kept;
        }
    for (size_t b = 0; b < 65536; ++b) start[b + 1] += start[b];
    order.resize((size_t)kept);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t k = 0; k < nlabels; ++k)
            if (index_of[k] >= 0) order[(size_t)fill[(size_t)key_of(k)]++] = (int32_t)k;
    }
    std::vector<int> used;
    for (int b = 0; b < 65536; ++b)
        if (start[(size_t)b + 1] > start[(size_t)b] + 1) used.push_back(b);
    parallel_for((int)used.size(), threads, [&](int u) {
        const int b = used[(size_t)u];
        std::sort(order.begin() + start[(size_t)b], order.begin() + start[(size_t)b + 1],
                  [&](int32_t x, int32_t y) { return name_of(x) < name_of(y); });
    });
    // lines: ranges of the sorted order on the threads, sizes first
    const int parts = threads > 1 ? 4 * threads : 1;
    std::vector<int64_t> part_bytes((size_t)parts + 1, 0);
    std::vector<std::vector<char>> chunks((size_t)parts);
    parallel_for(parts, threads, [&](int q) {
        const int64_t lo = kept * q / parts, hi = kept * (q + 1) / parts;
        std::vector<char> &buf = chunks[(size_t)q];
        buf.reserve((size_t)(hi - lo) * 40);
        char num[40];
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t k = order[(size_t)r];
            const std::string_view nm = name_of(k);
            const int nn = python_repr(values[index_of[k]], num);
            const size_t need = plen + nm.size() + 5 + (size_t)nn + 1;
            const size_t at = buf.size();
            buf.resize(at + need);
            char *w = buf.data() + at;
            memcpy(w, prefix, plen); w += plen;
            memcpy(w, nm.data(), nm.size()); w += nm.size();
            memcpy(w, ") \t= ", 5); w += 5;
            memcpy(w, num, (size_t)nn); w += nn;
            *w = '\n';
        }
        part_bytes[(size_t)q + 1] = (int64_t)buf.size();
    });
    for (int q = 0; q < parts; ++q) part_bytes[(size_t)q + 1] += part_bytes[(size_t)q];
    const int64_t total = part_bytes[(size_t)parts];
    char *res = static_cast<char *>(malloc((size_t)total + 8));
    if (!res) return CSV_NO_MEMORY;
    parallel_for(parts, threads, [&](int q) {
        if (!chunks[(size_t)q].empty()) memcpy(res + part_bytes[(size_t)q], chunks[(size_t)q].data(), chunks[(size_t)q].size());
    });
    *out = res;
    *out_len = total ? total - 1 : 0;  // without the last separator
    return CSV_OK;
}

void nodal_csv_free_buffer(char *p) { free(p); }

}  // extern "C"
