// dbreader.cpp -- BLAST database (format version 4, nucleotide) volume reader: alias files,
// index (.nin) and sequence (.nsq) files, ambiguity runs; loads subject ranges into an HBM shard.
//
// Formats: objtools/blast/seqdb_reader/index_files.txt:62-120 (index file), sequence_files.txt:60-170
// (2-bit data with the remainder count in the last byte, old/new ambiguity segments),
// alias_files.txt (key/value alias files).  Replaces, for this engine, what the reference reaches
// through API/seqsrc_seqdb.cpp:283-382 (s_SeqDbGetSequence & co) -> CSeqDB / seqdbvol.cpp.
// Host only; no third-party code.
#include "gbn_host.hpp"
#include "gbn_guard.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <sstream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Mapped {                             // read-only file mapping; the descriptor stays open for bulk reads (read_at)
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        n = (size_t)st.st_size;
        if (n) { void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); p = (m == MAP_FAILED) ? nullptr : (const uint8_t *)m; }
        if (n && !p) { ::close(fd); fd = -1; }
        return n == 0 || p != nullptr;
    }
    // bytes [off, off + k) into dst with pread: a shard's worth of sequence bytes read through the mapping costs a page fault
    // per 4 KB (3 M faults for 12.5 GB: the load's CPU time); the kernel's copy out of the page cache does not
    bool read_at(size_t off, uint8_t *dst, size_t k) const {
        if (off + k > n) return false;
        while (k) {
            const ssize_t r = fd >= 0 ? ::pread(fd, dst, k, (off_t)off) : -1;
            if (r <= 0) { if (!p) return false; std::memcpy(dst, p + off, k); return true; }
            dst += r; off += (size_t)r; k -= (size_t)r;
        }
        return true;
    }
    void close() { if (p) munmap((void *)p, n); if (fd >= 0) ::close(fd); p = nullptr; n = 0; fd = -1; }
};

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Volume {
    std::string name, title, date;
    Mapped nin, nsq;
    int32_t noids = 0, max_len = 0, first_oid = 0;
    int64_t vol_len = 0;
    const uint8_t *seq_arr = nullptr, *amb_arr = nullptr;       // big-endian Int4[noids + 1]
    uint32_t seq_off(int32_t i) const { return be32(seq_arr + 4 * (size_t)i); }
    uint32_t amb_off(int32_t i) const { return be32(amb_arr + 4 * (size_t)i); }
    int32_t length(int32_t i) const {
        const uint32_t a = seq_off(i), e = amb_off(i);
        if (e <= a || e > nsq.n) return -1;
        return (int32_t)((e - a - 1) * 4 + (nsq.p[e - 1] & 3));    // last byte: remainder count in its low 2 bits
    }
};

bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
std::string dir_of(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string() : p.substr(0, k + 1); }

}  // namespace

struct GbnBlastDb {
    std::vector<Volume> vols;
    std::string title;
    int64_t total_len = 0, alias_len = -1;
    int32_t total_seqs = 0, alias_nseq = -1, max_len = 0;
    ~GbnBlastDb() { for (auto &v : vols) { v.nin.close(); v.nsq.close(); } }
    const Volume *find(int32_t oid, int32_t &local) const {
        if (oid < 0 || oid >= total_seqs) return nullptr;
        size_t lo = 0, hi = vols.size();
        while (hi - lo > 1) { size_t m = (lo + hi) / 2; if (vols[m].first_oid > oid) hi = m; else lo = m; }
        local = oid - vols[lo].first_oid;
        return &vols[lo];
    }
};

namespace {

int open_volume(GbnBlastDb &db, const std::string &name) {
    Volume v; v.name = name;
    if (!v.nin.open(name + ".nin")) { gbn::set_error("cannot open " + name + ".nin"); return GBN_ERR_ARG; }
    if (!v.nsq.open(name + ".nsq")) { v.nin.close(); gbn::set_error("cannot open " + name + ".nsq"); return GBN_ERR_ARG; }
    const uint8_t *p = v.nin.p; const size_t n = v.nin.n;
    auto bad = [&](const char *why) { v.nin.close(); v.nsq.close(); gbn::set_error(name + ".nin: " + why); return GBN_ERR_ARG; };
    size_t off = 0;
    auto need = [&](size_t k) { return off + k <= n; };
    if (!need(8)) return bad("truncated header");
    if (be32(p) != 4) return bad("format version is not 4");
    if (be32(p + 4) != 0) return bad("not a nucleotide volume");
    off = 8;
    for (std::string *s : {&v.title, &v.date}) {                // length-prefixed strings
        if (!need(4)) return bad("truncated string");
        uint32_t l = be32(p + off); off += 4;
        if (!need(l)) return bad("truncated string");
        s->assign((const char *)p + off, l); off += l;
        while (!s->empty() && s->back() == '\0') s->pop_back();
    }
    if (!need(16)) return bad("truncated counts");
    v.noids = (int32_t)be32(p + off); off += 4;
    uint64_t len = 0; for (int i = 7; i >= 0; i--) len = (len << 8) | p[off + i];       // the one little-endian field
    v.vol_len = (int64_t)len; off += 8;
    v.max_len = (int32_t)be32(p + off); off += 4;
    if (v.noids < 0) return bad("negative sequence count");
    const size_t arr = 4 * ((size_t)v.noids + 1);
    if (!need(3 * arr)) return bad("offset arrays truncated");
    v.seq_arr = p + off + arr; v.amb_arr = p + off + 2 * arr;    // header-array first, then sequence-, ambig-array
    if (v.noids && (v.amb_off(v.noids - 1) > v.nsq.n || v.seq_off(v.noids) > v.nsq.n)) return bad("offsets beyond the .nsq file");
    v.first_oid = db.total_seqs;
    db.total_seqs += v.noids; db.total_len += v.vol_len; db.max_len = std::max(db.max_len, v.max_len);
    db.vols.push_back(std::move(v));
    return GBN_OK;
}

int open_name(GbnBlastDb &db, const std::string &name, int depth, bool top);

int open_alias(GbnBlastDb &db, const std::string &name, int depth, bool top) {
    if (depth > 8) { gbn::set_error("alias files nested too deeply: " + name); return GBN_ERR_ARG; }
    std::ifstream f(name + ".nal");
    if (!f) { gbn::set_error("cannot open " + name + ".nal"); return GBN_ERR_ARG; }
    std::string line, dblist;
    while (std::getline(f, line)) {
        size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos || line[a] == '#') continue;
        size_t b = line.find_first_of(" \t", a);
        std::string key = line.substr(a, b == std::string::npos ? std::string::npos : b - a), val;
        if (b != std::string::npos) { size_t c = line.find_first_not_of(" \t", b); if (c != std::string::npos) val = line.substr(c); }
        while (!val.empty() && (val.back() == '\r' || val.back() == ' ' || val.back() == '\t')) val.pop_back();
        if (key == "DBLIST") dblist = val;
        else if (key == "TITLE") { if (top) db.title = val; }
        else if (key == "NSEQ") { if (top) db.alias_nseq = (int32_t)atoll(val.c_str()); }
        else if (key == "LENGTH") { if (top) db.alias_len = atoll(val.c_str()); }
        else if (key == "GILIST" || key == "TILIST" || key == "SEQIDLIST" || key == "OIDLIST" || key == "MEMB_BIT" ||
                 key == "FIRST_OID" || key == "LAST_OID" || key == "MASKLIST") {
            gbn::set_error(name + ".nal: sequence filtering (" + key + ") is not supported"); return GBN_ERR_UNSUPPORTED;
        }                                                       // unknown keys are ignored (forward compatibility)
    }
    if (dblist.empty()) { gbn::set_error(name + ".nal has no DBLIST"); return GBN_ERR_ARG; }
    const std::string dir = dir_of(name);
    std::istringstream ss(dblist); std::string item;
    while (ss >> item) {
        if (item.size() >= 2 && item.front() == '"' && item.back() == '"') item = item.substr(1, item.size() - 2);
        int rc = open_name(db, item[0] == '/' ? item : dir + item, depth + 1, false);
        if (rc) return rc;
    }
    return GBN_OK;
}

int open_name(GbnBlastDb &db, const std::string &name, int depth, bool top) {
    if (file_exists(name + ".nal")) return open_alias(db, name, depth, top);    // an alias hides a volume of the same name
    return open_volume(db, name);
}

const uint8_t kNa4ToBlastna[16] = {15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14};    // CORE/blast_encoding.c:42-59

}  // namespace

extern "C" {

int gbn_blastdb_open(GbnBlastDb **out, const char *name) {
    return gbn::guard(__func__, [&]() -> int {
    if (!out || !name) { gbn::set_error("gbn_blastdb_open: bad argument"); return GBN_ERR_ARG; }
    auto *db = new GbnBlastDb();
    int rc = open_name(*db, name, 0, true);
    if (rc) { delete db; return rc; }
    if (db->title.empty() && !db->vols.empty()) db->title = db->vols[0].title;
    *out = db;
    return GBN_OK;
    });
}

void gbn_blastdb_close(GbnBlastDb *db) { delete db; }
int32_t gbn_blastdb_num_volumes(const GbnBlastDb *db) { return (int32_t)db->vols.size(); }
int32_t gbn_blastdb_num_seqs(const GbnBlastDb *db) { return db->total_seqs; }
// the statistics' database size: NSEQ / LENGTH of the alias file override the sums (alias_files.txt)
int32_t gbn_blastdb_stat_num_seqs(const GbnBlastDb *db) { return db->alias_nseq >= 0 ? db->alias_nseq : db->total_seqs; }
int64_t gbn_blastdb_stat_length(const GbnBlastDb *db) { return db->alias_len >= 0 ? db->alias_len : db->total_len; }
int64_t gbn_blastdb_total_length(const GbnBlastDb *db) { return db->total_len; }
int32_t gbn_blastdb_max_length(const GbnBlastDb *db) { return db->max_len; }
const char *gbn_blastdb_title(const GbnBlastDb *db) { return db->title.c_str(); }

int gbn_blastdb_volume_range(const GbnBlastDb *db, int32_t vol, int32_t *first_oid, int32_t *num_oids) {
    return gbn::guard(__func__, [&]() -> int {
    if (!db || vol < 0 || vol >= (int32_t)db->vols.size()) { gbn::set_error("volume index out of range"); return GBN_ERR_ARG; }
    if (first_oid) *first_oid = db->vols[vol].first_oid;
    if (num_oids) *num_oids = db->vols[vol].noids;
    return GBN_OK;
    });
}

int32_t gbn_blastdb_seq_length(const GbnBlastDb *db, int32_t oid) {
    return gbn::guard_as<int32_t>(__func__, (int32_t)-1, (int32_t)-1, [&]() -> int32_t {
    int32_t local; const Volume *v = db->find(oid, local);
    return v ? v->length(local) : -1;
    });
}

// ceil(len / 4) bytes of NCBI2na; the bits behind the last base are zero
int gbn_blastdb_get_ncbi2na(const GbnBlastDb *db, int32_t oid, uint8_t *dst, int64_t dst_bytes) {
    return gbn::guard(__func__, [&]() -> int {
    int32_t local; const Volume *v = db->find(oid, local);
    const int32_t len = v ? v->length(local) : -1;
    if (len < 0) { gbn::set_error("oid out of range or corrupt offsets"); return GBN_ERR_ARG; }
    const int64_t nb = ((int64_t)len + 3) / 4;
    if (dst_bytes < nb) { gbn::set_error("destination too small"); return GBN_ERR_ARG; }
    const uint8_t *src = v->nsq.p + v->seq_off(local);
    std::memcpy(dst, src, (size_t)nb);
    if (len & 3) dst[nb - 1] &= (uint8_t)(0xff << (2 * (4 - (len & 3))));
    return GBN_OK;
    });
}

// ambiguity runs of a sequence, decoded from either on-disk format (sequence_files.txt:131-170)
int32_t gbn_blastdb_num_ambiguities(const GbnBlastDb *db, int32_t oid) {
    return gbn::guard_as<int32_t>(__func__, (int32_t)-1, (int32_t)-1, [&]() -> int32_t {
    int32_t local; const Volume *v = db->find(oid, local);
    if (!v) return -1;
    const uint32_t a = v->amb_off(local), e = v->seq_off(local + 1);
    if (e < a + 4 || e > v->nsq.n) return 0;
    return (int32_t)(be32(v->nsq.p + a) & 0x7fffffffu);
    });
}

int gbn_blastdb_get_ambiguities(const GbnBlastDb *db, int32_t oid, int32_t *start, int32_t *length, uint8_t *na4, int32_t cap) {
    return gbn::guard(__func__, [&]() -> int {
    int32_t local; const Volume *v = db->find(oid, local);
    if (!v) { gbn::set_error("oid out of range"); return GBN_ERR_ARG; }
    const uint32_t a = v->amb_off(local), e = v->seq_off(local + 1);
    if (e < a + 4 || e > v->nsq.n) return GBN_OK;
    const uint32_t head = be32(v->nsq.p + a); const bool wide = (head >> 31) != 0; const uint32_t n = head & 0x7fffffffu;
    if ((uint64_t)a + 4 + (uint64_t)n * (wide ? 8 : 4) > e) { gbn::set_error("ambiguity data truncated"); return GBN_ERR_ARG; }
    for (uint32_t i = 0; i < n && (int32_t)i < cap; i++) {
        const uint8_t *s = v->nsq.p + a + 4 + (size_t)i * (wide ? 8 : 4);
        const uint32_t w0 = be32(s);
        if (wide) { na4[i] = (uint8_t)(w0 >> 28); length[i] = (int32_t)((w0 >> 16) & 0xfff) + 1; start[i] = (int32_t)be32(s + 4); }
        else { na4[i] = (uint8_t)(w0 >> 28); length[i] = (int32_t)((w0 >> 24) & 0xf) + 1; start[i] = (int32_t)(w0 & 0xffffff); }
    }
    return GBN_OK;
    });
}

// one BLASTNA code per base with the ambiguities applied (the traceback stage's encoding,
// eBlastEncodingNucleotide); sentinels != 0 puts the code 15 in front and behind
int gbn_blastdb_get_blastna(const GbnBlastDb *db, int32_t oid, uint8_t *dst, int64_t dst_bytes, int sentinels) {
    return gbn::guard(__func__, [&]() -> int {
    int32_t local; const Volume *v = db->find(oid, local);
    const int32_t len = v ? v->length(local) : -1;
    if (len < 0) { gbn::set_error("oid out of range or corrupt offsets"); return GBN_ERR_ARG; }
    if (dst_bytes < (int64_t)len + (sentinels ? 2 : 0)) { gbn::set_error("destination too small"); return GBN_ERR_ARG; }
    uint8_t *d = dst + (sentinels ? 1 : 0);
    const uint8_t *src = v->nsq.p + v->seq_off(local);
    for (int32_t i = 0; i < len; i++) d[i] = (src[i >> 2] >> (6 - 2 * (i & 3))) & 3;
    const int32_t na = gbn_blastdb_num_ambiguities(db, oid);
    if (na > 0) {
        std::vector<int32_t> st((size_t)na), ln((size_t)na); std::vector<uint8_t> val((size_t)na);
        int rc = gbn_blastdb_get_ambiguities(db, oid, st.data(), ln.data(), val.data(), na);
        if (rc) return rc;
        for (int32_t k = 0; k < na; k++)
            for (int32_t i = st[k]; i < st[k] + ln[k] && i < len; i++) d[i] = kNa4ToBlastna[val[k] & 15];
    }
    if (sentinels) { dst[0] = 15; dst[len + 1] = 15; }
    return GBN_OK;
    });
}

// subjects [first_oid, first_oid + num_oids) -> one slab (16-byte aligned subjects, 16 bytes in front,
// 128 behind) -> resident shard whose global OIDs start at first_oid
int gbn_blastdb_load_shard(const GbnBlastDb *db, int32_t first_oid, int32_t num_oids, GbnDb **out) {
    return gbn::guard(__func__, [&]() -> int {
    if (!db || !out || first_oid < 0 || num_oids < 0 || first_oid + (int64_t)num_oids > db->total_seqs) {
        gbn::set_error("gbn_blastdb_load_shard: bad oid range"); return GBN_ERR_ARG;
    }
    std::vector<int64_t> off((size_t)num_oids); std::vector<int32_t> len((size_t)num_oids);
    int64_t pos = 16;
    for (int32_t i = 0; i < num_oids; i++) {
        len[i] = gbn_blastdb_seq_length(db, first_oid + i);
        if (len[i] < 0) { gbn::set_error("corrupt sequence offsets"); return GBN_ERR_ARG; }
        off[i] = pos;
        pos += (((int64_t)len[i] + 3) / 4 + 15) / 16 * 16;
    }
    const int64_t nbytes = pos + 128;
    // the volumes' mapped .nsq bytes go to the device piece by piece through the library's pinned staging buffers, fills
    // (worker threads) overlapping uploads; a shard with a sequence beyond MAX_DBSEQ_LEN takes the one-slab form
    struct Ctx { const GbnBlastDb *db; int32_t first_oid; const int64_t *off; const int32_t *len; } cx{db, first_oid, off.data(), len.data()};
    auto fill = [](void *c, int32_t first, int32_t count, uint8_t *dst, int64_t base) -> int {
        const Ctx &x = *static_cast<const Ctx *>(c);
        for (int32_t i = first; i < first + count; i++) {
            int32_t local; const Volume *v = x.db->find(x.first_oid + i, local);
            const int64_t nb = ((int64_t)x.len[i] + 3) / 4;
            uint8_t *d = dst + (x.off[i] - base);
            if (!v || !v->nsq.read_at(v->seq_off(local), d, (size_t)nb)) { gbn::set_error("reading a sequence of the shard failed"); return GBN_ERR_ARG; }
            if (x.len[i] & 3) d[nb - 1] &= (uint8_t)(0xff << (2 * (4 - (x.len[i] & 3))));       // (the remainder count in the last byte's low bits)
        }
        return GBN_OK;
    };
    const bool one_slab = std::getenv("GBN_LOAD_ONE_SLAB") != nullptr;        // (A/B: the loader of rounds 1-5)
    int rc = (num_oids > 0 && !one_slab) ? gbn_db_new_streamed(out, nbytes, num_oids, off.data(), len.data(), first_oid, fill, &cx, 0) : GBN_ERR_UNSUPPORTED;
    if (rc == GBN_ERR_UNSUPPORTED) {
        std::vector<uint8_t> slab((size_t)nbytes, 0);
        for (int32_t i = 0; i < num_oids; i++) {
            rc = gbn_blastdb_get_ncbi2na(db, first_oid + i, slab.data() + off[i], ((int64_t)len[i] + 3) / 4);
            if (rc) return rc;
        }
        rc = gbn_db_new(out, slab.data(), nbytes, num_oids, off.data(), len.data(), first_oid, 0);
    }
    if (rc) return rc;
    // the ambiguity runs travel with the shard: the traceback stage needs the codes the 2-bit data cannot hold
    std::vector<int32_t> st, ln; std::vector<uint8_t> val;
    for (int32_t i = 0; i < num_oids && !rc; i++) {
        const int32_t k = gbn_blastdb_num_ambiguities(db, first_oid + i);
        if (k <= 0) continue;
        st.resize((size_t)k); ln.resize((size_t)k); val.resize((size_t)k);
        rc = gbn_blastdb_get_ambiguities(db, first_oid + i, st.data(), ln.data(), val.data(), k);
        if (!rc) rc = gbn_db_set_ambiguities(*out, i, k, st.data(), ln.data(), val.data());
    }
    if (rc) { gbn_db_free(*out); *out = nullptr; }
    return rc;
    });
}

}  // extern "C"
