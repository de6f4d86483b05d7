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
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string_view>
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

template <class T>
static T *dup(const std::vector<T> &v) {
    T *p = static_cast<T *>(malloc(v.size() * sizeof(T) + 8));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

extern "C" {

int nodal_csv_parse(const char *buf, int64_t len, nodal_csv_result *out) {
    memset(out, 0, sizeof *out);
    std::vector<int64_t> line_off;
    std::vector<int32_t> line_len, acode, bcode;
    std::vector<uint8_t> type_idx, nfields;
    std::vector<double> value;
    const size_t guess = (size_t)(len / 24 + 16);
    line_off.reserve(guess); line_len.reserve(guess); acode.reserve(guess); bcode.reserve(guess);
    type_idx.reserve(guess); nfields.reserve(guess); value.reserve(guess);
    StringIds name_ids(guess), node_ids(guess / 2);
    auto fail = [&](int status, int64_t line) {
        out->status = status;
        out->bad_line = line;
        return status;
    };

    // Two phases per block of rows: tokenize BLOCK rows, hashing their name and lead labels and
    // requesting the table slots; then run the table operations in file order (first-appearance ids,
    // first duplicate reported).  Errors found while tokenizing a later row of the block are only
    // returned after the earlier rows' table operations, so the FIRST irregular line is reported.
    constexpr int BLOCK = 16;
    struct Staged { std::string_view name, a, b; uint64_t hn, ha, hb; int64_t line; };
    Staged block[BLOCK];
    int staged = 0;
    auto flush = [&]() -> int {
        for (int q = 0; q < staged; ++q) {
            const Staged &st = block[q];
            bool fresh = false;
            name_ids.get(st.name, st.hn, &fresh);
            if (!fresh) return fail(CSV_DUPLICATE_NAME, st.line);
            acode.push_back(node_ids.get(st.a, st.ha, &fresh));  // first-appearance ids
            bcode.push_back(node_ids.get(st.b, st.hb, &fresh));
        }
        staged = 0;
        return CSV_OK;
    };
    // (an irregular row stops the file: the rows staged before it are checked for duplicates first)
    auto fail_after_flush = [&](int status, int64_t line) {
        const int st_ = flush();
        return st_ != CSV_OK ? st_ : fail(status, line);
    };

    int64_t pos = 0, lineno = 0;
    char numbuf[64];
    while (pos < len) {
        const char *nl = static_cast<const char *>(memchr(buf + pos, '\n', (size_t)(len - pos)));
        int64_t end = nl ? nl - buf : len;
        const int64_t next = nl ? end + 1 : len;
        const int64_t start = pos;
        pos = next;
        const int64_t this_line = lineno++;
        if (end > start && buf[end - 1] == '\r') --end;
        if (end == start) continue;  // empty line: csv.reader yields [] and the reference skips it
        bool blank = true;
        for (int64_t i = start; i < end; ++i) {
            const char ch = buf[i];
            if (ch == '"') return fail_after_flush(CSV_QUOTES, this_line);
            if (ch == '\r') return fail_after_flush(CSV_NEWLINE_IN_FIELD, this_line);
            if (ch != ' ' && ch != '\t') blank = false;
        }
        if (blank) return fail_after_flush(CSV_BLANK_WITH_SPACES, this_line);  // the reference raises IndexError there
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
        if (f[0].empty()) return fail_after_flush(CSV_EMPTY_FIRST_FIELD, this_line);
        if (f[0][0] == '#') continue;  // comment row
        if (nf > 8) return fail_after_flush(CSV_TOO_MANY_FIELDS, this_line);
        if (nf < 2) return fail_after_flush(CSV_FIELD_COUNT, this_line);
        const int ti = type_index(f[1]);
        if (ti < 0) return fail_after_flush(CSV_UNKNOWN_TYPE, this_line);
        if (nf != TYPE_FIELDS[ti]) return fail_after_flush(CSV_FIELD_COUNT, this_line);
        // value: plain decimal spellings only; everything float() accepts beyond that
        // ("1_0", " 1 ", "nan", "inf") goes to the exact parser
        const std::string_view v = f[2];
        if (v.empty() || v.size() >= sizeof numbuf) return fail_after_flush(CSV_BAD_VALUE, this_line);
        bool digit = false;
        for (char ch : v) {
            if (ch >= '0' && ch <= '9') digit = true;
            else if (ch != '+' && ch != '-' && ch != '.' && ch != 'e' && ch != 'E')
                return fail_after_flush(CSV_BAD_VALUE, this_line);
        }
        if (!digit) return fail_after_flush(CSV_BAD_VALUE, this_line);
        double val;
        if (!parse_plain_decimal(v, &val)) {  // exponent forms, long mantissas: strtod (correctly rounded too)
            memcpy(numbuf, v.data(), v.size());
            numbuf[v.size()] = 0;
            char *endp = nullptr;
            errno = 0;
            val = strtod(numbuf, &endp);
            if (endp != numbuf + v.size()) return fail_after_flush(CSV_BAD_VALUE, this_line);
        }
        Staged &st = block[staged++];
        st.name = f[0]; st.a = f[3]; st.b = f[4];
        st.hn = StringIds::hash(f[0]); st.ha = StringIds::hash(f[3]); st.hb = StringIds::hash(f[4]);
        st.line = this_line;
        name_ids.prefetch(st.hn);
        node_ids.prefetch(st.ha);
        node_ids.prefetch(st.hb);
        line_off.push_back(start);
        line_len.push_back((int32_t)(end - start));
        type_idx.push_back((uint8_t)ti);
        nfields.push_back((uint8_t)nf);
        value.push_back(val);
        if (staged == BLOCK) {
            const int st_ = flush();
            if (st_ != CSV_OK) return st_;
        }
    }
    {
        const int st_ = flush();
        if (st_ != CSV_OK) return st_;
    }
    const std::vector<std::string_view> &names = name_ids.items, &labels = node_ids.items;
    if (names.empty()) return fail(CSV_NO_COMPONENTS, 0);

    auto join = [](const std::vector<std::string_view> &v, int64_t *bytes) -> char * {
        size_t total = 0;
        for (const auto &s : v) total += s.size() + 1;
        char *p = static_cast<char *>(malloc(total + 8));
        if (!p) return nullptr;
        size_t o = 0;
        for (const auto &s : v) {
            memcpy(p + o, s.data(), s.size());
            o += s.size();
            p[o++] = '\n';
        }
        *bytes = (int64_t)(total ? total - 1 : 0);  // without the last separator
        return p;
    };
    out->nrows = (int64_t)names.size();
    out->nnodes = (int64_t)labels.size();
    out->line_off = dup(line_off);
    out->line_len = dup(line_len);
    out->type_idx = dup(type_idx);
    out->nfields = dup(nfields);
    out->value = dup(value);
    out->acode = dup(acode);
    out->bcode = dup(bcode);
    out->names_blob = join(names, &out->names_bytes);
    out->labels_blob = join(labels, &out->labels_bytes);
    if (!out->line_off || !out->line_len || !out->type_idx || !out->nfields || !out->value || !out->acode ||
        !out->bcode || !out->names_blob || !out->labels_blob) {
        nodal_csv_free(out);
        out->status = CSV_NO_MEMORY;
        return CSV_NO_MEMORY;
    }
    return CSV_OK;
}

}  // extern "C"
