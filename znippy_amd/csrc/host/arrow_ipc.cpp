#include "arrow_ipc.h"

#include <algorithm>
#include <cstring>

#include "fb.h"

namespace aipc {

namespace {
constexpr size_t ALIGN = 64;  // arrow-rs IpcWriteOptions::default().alignment
// Message.fbs / Schema.fbs constants
constexpr int16_t METADATA_V5 = 4;
constexpr uint8_t HDR_SCHEMA = 1, HDR_RECORD_BATCH = 3;
constexpr uint8_t T_INT = 2, T_UTF8 = 5, T_BOOL = 6, T_FSB = 15;

size_t pad_to(size_t n, size_t a) { return (n + a - 1) / a * a; }

struct Ser {
    std::vector<uint8_t> out;
    void bytes(const void *p, size_t n) { out.insert(out.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
    void zeros(size_t n) { out.insert(out.end(), n, 0); }
    void message(const fb::Builder &b, const std::vector<uint8_t> &body) {
        const uint32_t cont = 0xFFFFFFFFu;
        const size_t fbsz = b.size();
        const size_t aligned = pad_to(fbsz + 8, ALIGN);
        const int32_t meta_len = (int32_t)(aligned - 8);
        bytes(&cont, 4);
        bytes(&meta_len, 4);
        bytes(b.data(), fbsz);
        zeros(aligned - 8 - fbsz);
        bytes(body.data(), body.size());
    }
    void eos() {
        const uint32_t cont = 0xFFFFFFFFu, zero = 0;
        bytes(&cont, 4);
        bytes(&zero, 4);
    }
};

uint32_t build_type(fb::Builder &b, Kind k, uint8_t *type_type) {
    switch (k) {
    case Kind::Utf8:
        *type_type = T_UTF8;
        b.start_table(0);
        return b.end_table();
    case Kind::Bool:
        *type_type = T_BOOL;
        b.start_table(0);
        return b.end_table();
    case Kind::FixedBin32:
        *type_type = T_FSB;
        b.start_table(1);
        b.add_scalar<int32_t>(0, 32, 0);
        return b.end_table();
    case Kind::UInt32:
    case Kind::UInt64:
    case Kind::Int8: {
        *type_type = T_INT;
        b.start_table(2);
        b.add_scalar<int32_t>(0, k == Kind::UInt32 ? 32 : (k == Kind::UInt64 ? 64 : 8), 0);
        b.add_scalar<uint8_t>(1, k == Kind::Int8 ? 1 : 0, 0);
        return b.end_table();
    }
    }
    return 0;
}

uint32_t build_kv(fb::Builder &b, const std::map<std::string, std::string> &md) {
    if (md.empty()) return 0;
    std::vector<uint32_t> kvs;
    for (const auto &kv : md) {  // std::map iterates in sorted key order
        uint32_t k = b.create_string(kv.first), v = b.create_string(kv.second);
        b.start_table(2);
        b.add_offset(0, k);
        b.add_offset(1, v);
        kvs.push_back(b.end_table());
    }
    return b.create_offset_vector(kvs);
}

void schema_message(Ser &s, const Batch &schema_of, const std::map<std::string, std::string> &md) {
    fb::Builder b;
    std::vector<uint32_t> fields;
    for (const Column &c : schema_of.cols) {
        uint32_t name = b.create_string(c.name);
        uint8_t tt = 0;
        uint32_t type = build_type(b, c.kind, &tt);
        uint32_t children = b.create_offset_vector({});
        b.start_table(7);
        b.add_offset(0, name);
        b.add_scalar<uint8_t>(1, c.nullable ? 1 : 0, 0);
        b.add_scalar<uint8_t>(2, tt, 0);
        b.add_offset(3, type);
        b.add_offset(5, children);
        fields.push_back(b.end_table());
    }
    uint32_t fvec = b.create_offset_vector(fields);
    uint32_t kv = build_kv(b, md);
    b.start_table(4);
    b.add_scalar<int16_t>(0, 0, 0);  // little endian (default)
    b.add_offset(1, fvec);
    b.add_offset(2, kv);
    uint32_t schema = b.end_table();
    b.start_table(5);
    b.add_scalar<int16_t>(0, METADATA_V5, 0);
    b.add_scalar<uint8_t>(1, HDR_SCHEMA, 0);
    b.add_offset(2, schema);
    b.add_scalar<int64_t>(3, 0, 0);
    uint32_t msg = b.end_table();
    b.finish(msg);
    s.message(b, {});
}

struct BufRec {
    int64_t offset, length;
};
struct NodeRec {
    int64_t length, null_count;
};

void add_buffer(std::vector<uint8_t> &body, std::vector<BufRec> &recs, const void *p, size_t n) {
    recs.push_back({(int64_t)body.size(), (int64_t)n});
    if (n) body.insert(body.end(), (const uint8_t *)p, (const uint8_t *)p + n);
    body.resize(pad_to(body.size(), ALIGN), 0);
}

void batch_message(Ser &s, const Batch &batch) {
    const size_t rows = batch.rows();
    std::vector<uint8_t> body;
    std::vector<BufRec> bufs;
    std::vector<NodeRec> nodes;
    for (const Column &c : batch.cols) {
        nodes.push_back({(int64_t)rows, 0});
        add_buffer(body, bufs, nullptr, 0);  // validity: absent (no nulls)
        switch (c.kind) {
        case Kind::Utf8: {
            std::vector<int32_t> offs(rows + 1, 0);
            std::string data;
            for (size_t i = 0; i < rows; i++) {
                data += c.str[i];
                offs[i + 1] = (int32_t)data.size();
            }
            add_buffer(body, bufs, offs.data(), offs.size() * 4);
            add_buffer(body, bufs, data.data(), data.size());
            break;
        }
        case Kind::UInt32: add_buffer(body, bufs, c.u32.data(), rows * 4); break;
        case Kind::UInt64: add_buffer(body, bufs, c.u64.data(), rows * 8); break;
        case Kind::Int8: add_buffer(body, bufs, c.u8.data(), rows); break;
        case Kind::FixedBin32: add_buffer(body, bufs, c.u8.data(), rows * 32); break;
        case Kind::Bool: {
            std::vector<uint8_t> bits((rows + 7) / 8, 0);
            for (size_t i = 0; i < rows; i++)
                if (c.u8[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
            add_buffer(body, bufs, bits.data(), bits.size());
            break;
        }
        }
    }
    fb::Builder b;
    uint32_t bvec = b.create_struct_vector(bufs.data(), bufs.size(), 16, 8);
    uint32_t nvec = b.create_struct_vector(nodes.data(), nodes.size(), 16, 8);
    b.start_table(5);
    b.add_scalar<int64_t>(0, (int64_t)rows, 0);
    b.add_offset(1, nvec);
    b.add_offset(2, bvec);
    uint32_t rb = b.end_table();
    b.start_table(5);
    b.add_scalar<int16_t>(0, METADATA_V5, 0);
    b.add_scalar<uint8_t>(1, HDR_RECORD_BATCH, 0);
    b.add_offset(2, rb);
    b.add_scalar<int64_t>(3, (int64_t)body.size(), 0);
    uint32_t msg = b.end_table();
    b.finish(msg);
    s.message(b, body);
}

bool kind_of(const fb::Table &field, Kind *k) {
    const uint8_t tt = field.scalar<uint8_t>(2, 0);
    const fb::Table ty = field.table(3);
    if (tt == T_UTF8) { *k = Kind::Utf8; return true; }
    if (tt == T_BOOL) { *k = Kind::Bool; return true; }
    if (tt == T_FSB) { *k = Kind::FixedBin32; return ty.ok() && ty.scalar<int32_t>(0, 0) == 32; }
    if (tt == T_INT && ty.ok()) {
        const int32_t bw = ty.scalar<int32_t>(0, 0);
        const bool sgn = ty.scalar<uint8_t>(1, 0) != 0;
        if (bw == 32 && !sgn) { *k = Kind::UInt32; return true; }
        if (bw == 64 && !sgn) { *k = Kind::UInt64; return true; }
        if (bw == 8 && sgn) { *k = Kind::Int8; return true; }
    }
    return false;
}
}  // namespace

size_t Column::rows() const {
    switch (kind) {
    case Kind::Utf8: return str.size();
    case Kind::UInt32: return u32.size();
    case Kind::UInt64: return u64.size();
    case Kind::FixedBin32: return u8.size() / 32;
    default: return u8.size();
    }
}

std::vector<uint8_t> write_stream(const std::vector<Batch> &batches, const Batch &schema_of,
                                  const std::map<std::string, std::string> &metadata) {
    Ser s;
    schema_message(s, schema_of, metadata);
    for (const Batch &b : batches) batch_message(s, b);
    s.eos();
    return std::move(s.out);
}

bool read_stream(const uint8_t *p, size_t n, Batch *out, std::map<std::string, std::string> *metadata,
                 std::string *err) {
    size_t pos = 0;
    bool have_schema = false;
    auto fail = [&](const char *m) { if (err) *err = m; return false; };
    while (pos + 4 <= n) {
        uint32_t w;
        std::memcpy(&w, p + pos, 4);
        int32_t meta_len;
        if (w == 0xFFFFFFFFu) {
            if (pos + 8 > n) return fail("truncated message prefix");
            std::memcpy(&meta_len, p + pos + 4, 4);
            pos += 8;
        } else {  // pre-0.15 framing: length only
            meta_len = (int32_t)w;
            pos += 4;
        }
        if (meta_len == 0) break;  // end of stream
        if (meta_len < 0 || pos + (size_t)meta_len > n) return fail("bad metadata length");
        const fb::Table msg = fb::root(p + pos, (size_t)meta_len);
        if (!msg.ok()) return fail("bad message flatbuffer");
        pos += (size_t)meta_len;
        const uint8_t htype = msg.scalar<uint8_t>(1, 0);
        const int64_t body_len = msg.scalar<int64_t>(3, 0);
        if (body_len < 0 || pos + (size_t)body_len > n) return fail("bad body length");
        const uint8_t *body = p + pos;
        pos += (size_t)body_len;
        const fb::Table hdr = msg.table(2);
        if (htype == HDR_SCHEMA) {
            if (!hdr.ok()) return fail("schema header missing");
            uint32_t nf;
            hdr.vec(1, &nf, 4);
            if (!have_schema && out->cols.empty()) {
                for (uint32_t i = 0; i < nf; i++) {
                    const fb::Table f = hdr.vec_table(1, i);
                    Column c;
                    c.name = f.str(0);
                    c.nullable = f.scalar<uint8_t>(1, 0) != 0;
                    if (!kind_of(f, &c.kind)) return fail("unsupported column type");
                    out->cols.push_back(std::move(c));
                }
            } else {
                if (nf != out->cols.size()) return fail("schema mismatch between streams");
                for (uint32_t i = 0; i < nf; i++) {
                    Kind k;
                    const fb::Table f = hdr.vec_table(1, i);
                    if (!kind_of(f, &k) || k != out->cols[i].kind || f.str(0) != out->cols[i].name)
                        return fail("schema mismatch between streams");
                }
            }
            if (metadata) {
                uint32_t nk;
                hdr.vec(2, &nk, 4);
                for (uint32_t i = 0; i < nk; i++) {
                    const fb::Table kv = hdr.vec_table(2, i);
                    (*metadata)[kv.str(0)] = kv.str(1);
                }
            }
            have_schema = true;
        } else if (htype == HDR_RECORD_BATCH) {
            if (!have_schema || !hdr.ok()) return fail("record batch before schema");
            const int64_t rows = hdr.scalar<int64_t>(0, 0);
            uint32_t nn, nb;
            const uint8_t *nodes = hdr.vec(1, &nn, 16);
            const uint8_t *bufs = hdr.vec(2, &nb, 16);
            if (hdr.field_off(3)) return fail("compressed record batches are not supported");
            if (rows < 0 || rows >= (int64_t)1 << 40 || nn != out->cols.size() || (nn && !nodes)) return fail("bad record batch");
            uint32_t bi = 0;
            auto buf = [&](const uint8_t **ptr, int64_t *len) -> bool {
                if (bi >= nb) return false;
                int64_t off;
                std::memcpy(&off, bufs + 16 * bi, 8);
                std::memcpy(len, bufs + 16 * bi + 8, 8);
                bi++;
                if (off < 0 || *len < 0 || off > body_len || *len > body_len - off) return false;
                *ptr = body + off;
                return true;
            };
            for (uint32_t ci = 0; ci < nn; ci++) {
                int64_t nulls;
                std::memcpy(&nulls, nodes + 16 * ci + 8, 8);
                if (nulls != 0) return fail("null values in a non-null column");
                Column &c = out->cols[ci];
                const uint8_t *ptr;
                int64_t len;
                if (!buf(&ptr, &len)) return fail("bad validity buffer");
                switch (c.kind) {
                case Kind::Utf8: {
                    const uint8_t *op, *dp;
                    int64_t ol, dl;
                    if (!buf(&op, &ol) || !buf(&dp, &dl)) return fail("bad utf8 buffers");
                    if (rows && (rows >= (int64_t)1 << 40 || ol < (rows + 1) * 4)) return fail("short offsets buffer");
                    for (int64_t i = 0; i < rows; i++) {
                        int32_t a, b;
                        std::memcpy(&a, op + 4 * i, 4);
                        std::memcpy(&b, op + 4 * i + 4, 4);
                        if (a < 0 || b < a || b > dl) return fail("bad utf8 offsets");
                        c.str.emplace_back((const char *)dp + a, (size_t)(b - a));
                    }
                    break;
                }
                case Kind::UInt32:
                    if (!buf(&ptr, &len) || len < rows * 4) return fail("bad u32 buffer");
                    c.u32.insert(c.u32.end(), (const uint32_t *)ptr, (const uint32_t *)ptr + rows);
                    break;
                case Kind::UInt64:
                    if (!buf(&ptr, &len) || len < rows * 8) return fail("bad u64 buffer");
                    { size_t o = c.u64.size(); c.u64.resize(o + rows); if (rows) std::memcpy(&c.u64[o], ptr, rows * 8); }
                    break;
                case Kind::Int8:
                    if (!buf(&ptr, &len) || len < rows) return fail("bad i8 buffer");
                    c.u8.insert(c.u8.end(), ptr, ptr + rows);
                    break;
                case Kind::FixedBin32:
                    if (!buf(&ptr, &len) || len < rows * 32) return fail("bad fixed-size-binary buffer");
                    c.u8.insert(c.u8.end(), ptr, ptr + rows * 32);
                    break;
                case Kind::Bool:
                    if (!buf(&ptr, &len) || len < (rows + 7) / 8) return fail("bad bool buffer");
                    for (int64_t i = 0; i < rows; i++) c.u8.push_back((ptr[i >> 3] >> (i & 7)) & 1);
                    break;
                }
            }
        }  // other message kinds (dictionary batches) are not produced by this format
    }
    if (!have_schema) return fail("no schema message");
    return true;
}

}  // namespace aipc
