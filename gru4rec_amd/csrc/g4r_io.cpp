// Host-side event-table loader.  Replaces, for TAB separated files, the pandas work in front of
// the hot path: `pd.read_csv(..., dtype={session: int32, item: str})` (run.py:45-78), `data[item_key].unique()` and the
// `itemidmap` lookup that produces ItemIdx (gru4rec.py:534-538).  The file is mmap'ed, cut into one chunk per thread at
// line boundaries, every thread parses its chunk and interns the item-id byte strings in a private table; the private
// tables are then merged in file order, so item indices come out in order of first appearance -- exactly the order of
// pandas' unique(), which fixes the row of every item in Wy / E.
//
// Files this parser does not cover (quoted fields, empty or non-integer session ids, missing values, session ids beyond
// int32) are reported as G4R_IO_UNSUPPORTED and the Python side falls back to pandas; nothing is guessed.
#include "../../include/gru4rec_hip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

void g4r_set_error(const char* fmt, ...);      // g4r_host_model.hpp

namespace {

struct Slice { const char* p; uint32_t len; };

inline uint64_t hash_bytes(const char* p, uint32_t n) {
    // 8 bytes at a time multiply-xorshift; item ids are short (typically < 16 bytes)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xff51afd7ed558ccdull);
    while (n >= 8) {
        uint64_t w; memcpy(&w, p, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32;
        p += 8; n -= 8;
    }
    if (n) {
        uint64_t w = 0; memcpy(&w, p, n);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
    }
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 32);
}

// open-addressing string -> dense id table; keys are slices of the mapped file
struct Interner {
    std::vector<int32_t> slot;       // -1 = empty, else id
    std::vector<uint32_t> tag;       // low hash bits of the occupant
    std::vector<Slice> keys;         // id -> bytes, in order of first appearance
    std::vector<uint64_t> hashes;    // id -> full hash (reused when merging / growing)
    uint64_t mask = 0;
    Interner() { rehash(1 << 12); }
    void rehash(size_t cap) {
        slot.assign(cap, -1); tag.assign(cap, 0); mask = cap - 1;
        for (size_t id = 0; id < keys.size(); ++id) place(hashes[id], (int32_t)id);
    }
    void place(uint64_t h, int32_t id) {
        uint64_t i = h & mask;
        while (slot[i] >= 0) i = (i + 1) & mask;
        slot[i] = id; tag[i] = (uint32_t)(h >> 32);
    }
    int32_t intern(const char* p, uint32_t n, uint64_t h) {
        const uint32_t t = (uint32_t)(h >> 32);
        uint64_t i = h & mask;
        while (slot[i] >= 0) {
            if (tag[i] == t) {
                const Slice& k = keys[slot[i]];
                if (k.len == n && memcmp(k.p, p, n) == 0) return slot[i];
            }
            i = (i + 1) & mask;
        }
        const int32_t id = (int32_t)keys.size();
        keys.push_back(Slice{p, n}); hashes.push_back(h);
        slot[i] = id; tag[i] = t;
        if (keys.size() * 2 > slot.size()) rehash(slot.size() * 4);
        return id;
    }
};

struct Chunk {
    const char *beg, *end;
    std::vector<int32_t> session, item;
    std::vector<int64_t> time_i;
    std::vector<std::pair<int64_t, double>> time_f;      // rows whose time field is not a plain integer
    Interner items;
    std::vector<int32_t> to_global;
    int unsupported = 0;
    int64_t row0 = 0;
};

// plain decimal integer (optional sign); false when the field is anything else
inline bool parse_int(const char* p, const char* e, int64_t* out) {
    if (p == e) return false;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = (*p == '-'); ++p; if (p == e) return false; }
    if (e - p > 18) return false;
    uint64_t v = 0;
    for (; p < e; ++p) {
        const unsigned d = (unsigned)(*p - '0');
        if (d > 9) return false;
        v = v * 10 + d;
    }
    *out = neg ? -(int64_t)v : (int64_t)v;
    return true;
}

void parse_chunk(Chunk& c, int cs, int ci, int ct, int ncol_needed) {
    const char* p = c.beg;
    const size_t guess = (size_t)(c.end - c.beg) / 24 + 16;
    c.session.reserve(guess); c.item.reserve(guess);
    if (ct >= 0) c.time_i.reserve(guess);
    while (p < c.end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(c.end - p));
        const char* le = nl ? nl : c.end;
        const char* next = nl ? nl + 1 : c.end;
        if (le > p && le[-1] == '\r') --le;
        if (le == p) { p = next; continue; }               // blank line (pandas: skip_blank_lines)
        const char *fs = nullptr, *fse = nullptr, *fi = nullptr, *fie = nullptr, *ft = nullptr, *fte = nullptr;
        int col = 0;
        const char* f = p;
        while (true) {
            const char* tab = (const char*)memchr(f, '\t', (size_t)(le - f));
            const char* fe = tab ? tab : le;
            if (col == cs) { fs = f; fse = fe; }
            if (col == ci) { fi = f; fie = fe; }
            if (col == ct) { ft = f; fte = fe; }
            ++col;
            if (!tab || col >= ncol_needed) break;
            f = tab + 1;
        }
        int64_t sv = 0;
        if (!fs || !fi || (ct >= 0 && !ft) || fi == fie || !parse_int(fs, fse, &sv) || sv > INT32_MAX || sv < INT32_MIN) {
            c.unsupported = 1;
            return;
        }
        if (ct >= 0) {
            int64_t tv = 0;
            if (!parse_int(ft, fte, &tv)) {
                if (ft == fte) { c.unsupported = 1; return; }
                std::string tmp(ft, fte);
                char* endp = nullptr;
                errno = 0;
                const double d = strtod(tmp.c_str(), &endp);
                if (endp != tmp.c_str() + tmp.size() || d != d) { c.unsupported = 1; return; }
                c.time_f.emplace_back((int64_t)c.session.size(), d);
            }
            c.time_i.push_back(tv);
        }
        const uint32_t n = (uint32_t)(fie - fi);
        c.session.push_back((int32_t)sv);
        c.item.push_back(c.items.intern(fi, n, hash_bytes(fi, n)));
        p = next;
    }
}

template <class F>
void run_parallel(int n, F&& fn) {
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back([&fn, i] { fn(i); });
    fn(0);
    for (auto& t : th) t.join();
}

}  // namespace

struct g4r_events {
    int64_t n_rows = 0, n_items = 0;
    int time_is_float = 0, has_time = 0;
    std::vector<Chunk> chunks;            // per-thread parse results; g4r_events_copy scatters them into the caller's arrays
    std::vector<int64_t> item_off;
    std::string item_bytes;
};

extern "C" {

int g4r_events_load(const char* path, const char* session_col, const char* item_col, const char* time_col, int32_t n_threads,
                    g4r_events** out) {
    if (!path || !session_col || !item_col || !out) { g4r_set_error("g4r_events_load: null argument"); return -1; }
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { g4r_set_error("g4r_events_load: cannot open %s: %s", path, strerror(errno)); return -2; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); g4r_set_error("g4r_events_load: fstat failed on %s", path); return -2; }
    const size_t size = (size_t)sb.st_size;
    if (size == 0) { close(fd); g4r_set_error("g4r_events_load: %s is empty", path); return -3; }
    const char* base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { g4r_set_error("g4r_events_load: mmap failed on %s", path); return -2; }
    madvise((void*)base, size, MADV_WILLNEED);
    int rc = 0;
    g4r_events* ev = nullptr;
    do {
        // header
        const char* nl = (const char*)memchr(base, '\n', size);
        const char* he = nl ? nl : base + size;
        const char* body = nl ? nl + 1 : base + size;
        if (he > base && he[-1] == '\r') --he;
        int cs = -1, ci = -1, ct = -1, col = 0;
        for (const char* f = base; ; ++col) {
            const char* tab = (const char*)memchr(f, '\t', (size_t)(he - f));
            const char* fe = tab ? tab : he;
            const std::string name(f, fe);
            if (name == session_col && cs < 0) cs = col;
            if (name == item_col && ci < 0) ci = col;
            if (time_col && name == time_col && ct < 0) ct = col;
            if (!tab) break;
            f = tab + 1;
        }
        if (cs < 0 || ci < 0 || (time_col && ct < 0)) {
            g4r_set_error("g4r_events_load: column %s not in the header of %s", cs < 0 ? session_col : ci < 0 ? item_col : time_col, path);
            rc = -4;
            break;
        }
        const int ncol_needed = std::max(cs, std::max(ci, ct)) + 1;
        // n_threads: 0 = all cores, > 0 = at most that many (never more than one per MiB of input), < 0 = exactly -n_threads
        int nt = n_threads > 0 ? n_threads : n_threads < 0 ? -n_threads : (int)std::thread::hardware_concurrency();
        nt = std::max(1, std::min(nt, 64));
        const size_t body_size = (size_t)(base + size - body);
        if (n_threads >= 0) nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, body_size / (1 << 20) + 1));
        ev = new g4r_events();
        std::vector<Chunk>& chunks = ev->chunks;
        chunks.resize(nt);
        const char* cur = body;
        for (int i = 0; i < nt; ++i) {
            const char* want = (i + 1 == nt) ? base + size : body + body_size * (size_t)(i + 1) / (size_t)nt;
            if (want < cur) want = cur;
            if (i + 1 < nt && want < base + size) {
                const char* e = (const char*)memchr(want, '\n', (size_t)(base + size - want));
                want = e ? e + 1 : base + size;
            }
            chunks[i].beg = cur; chunks[i].end = want;
            cur = want;
        }
        run_parallel(nt, [&](int i) {
            // quoting rules are left to pandas: any double quote in the chunk makes the file "unsupported"
            if (memchr(chunks[i].beg, '"', (size_t)(chunks[i].end - chunks[i].beg))) { chunks[i].unsupported = 1; return; }
            parse_chunk(chunks[i], cs, ci, ct, ncol_needed);
        });
        bool bad = memchr(base, '"', (size_t)(body - base)) != nullptr;
        for (auto& c : chunks) bad |= c.unsupported != 0;
        if (bad) { rc = G4R_IO_UNSUPPORTED; break; }
        // merge the private item tables in file order: global ids in order of first appearance
        Interner global;
        for (auto& c : chunks) {
            c.to_global.resize(c.items.keys.size());
            for (size_t id = 0; id < c.items.keys.size(); ++id)
                c.to_global[id] = global.intern(c.items.keys[id].p, c.items.keys[id].len, c.items.hashes[id]);
        }
        int64_t rows = 0;
        bool any_float = false;
        for (auto& c : chunks) { c.row0 = rows; rows += (int64_t)c.session.size(); any_float |= !c.time_f.empty(); }
        ev->n_rows = rows;
        ev->n_items = (int64_t)global.keys.size();
        ev->has_time = ct >= 0;
        ev->time_is_float = any_float;
        ev->item_off.resize((size_t)ev->n_items + 1);
        size_t total = 0;
        for (size_t id = 0; id < global.keys.size(); ++id) { ev->item_off[id] = (int64_t)total; total += global.keys[id].len; }
        ev->item_off[(size_t)ev->n_items] = (int64_t)total;
        ev->item_bytes.resize(total);
        for (size_t id = 0; id < global.keys.size(); ++id)
            memcpy(&ev->item_bytes[(size_t)ev->item_off[id]], global.keys[id].p, global.keys[id].len);
        for (auto& c : chunks) { c.items = Interner(); c.beg = c.end = nullptr; }      // the slices die with the mapping
    } while (false);
    munmap((void*)base, size);
    if (rc != 0) { delete ev; return rc; }
    *out = ev;
    return 0;
}

int64_t g4r_events_rows(const g4r_events* ev) { return ev ? ev->n_rows : -1; }
int64_t g4r_events_items(const g4r_events* ev) { return ev ? ev->n_items : -1; }
int64_t g4r_events_item_bytes(const g4r_events* ev) { return ev ? (int64_t)ev->item_bytes.size() : -1; }
int32_t g4r_events_time_kind(const g4r_events* ev) { return !ev ? -1 : !ev->has_time ? 0 : ev->time_is_float ? 2 : 1; }

int g4r_events_copy(const g4r_events* ev, int32_t* session, int32_t* item_idx, void* time, int64_t* item_off, char* item_bytes) {
    if (!ev) { g4r_set_error("g4r_events_copy: null handle"); return -1; }
    const bool as_float = ev->time_is_float != 0;
    run_parallel((int)ev->chunks.size(), [&](int i) {
        const Chunk& c = ev->chunks[i];
        const size_t n = c.session.size();
        if (!n) return;
        if (session) memcpy(session + c.row0, c.session.data(), n * sizeof(int32_t));
        if (item_idx) {
            int32_t* dst = item_idx + c.row0;
            for (size_t r = 0; r < n; ++r) dst[r] = c.to_global[c.item[r]];
        }
        if (time && ev->has_time) {
            if (as_float) {
                double* t = (double*)time + c.row0;
                for (size_t r = 0; r < n; ++r) t[r] = (double)c.time_i[r];
                for (auto& kv : c.time_f) t[kv.first] = kv.second;
            } else {
                memcpy((int64_t*)time + c.row0, c.time_i.data(), n * sizeof(int64_t));
            }
        }
    });
    if (item_off) memcpy(item_off, ev->item_off.data(), ev->item_off.size() * sizeof(int64_t));
    if (item_bytes && !ev->item_bytes.empty()) memcpy(item_bytes, ev->item_bytes.data(), ev->item_bytes.size());
    return 0;
}

void g4r_events_free(g4r_events* ev) { delete ev; }

}  // extern "C"
