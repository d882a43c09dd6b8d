// Minimal FlatBuffers wire-format codec (builder + table reader) — just enough for Arrow IPC
// Message/Schema/RecordBatch metadata.  Written from the FlatBuffers binary-format description
// (tables with vtables, back-to-front construction, uoffset/soffset/voffset), no generated code.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace fb {

class Builder {
  public:
    explicit Builder(size_t initial = 1024) : buf_(initial), head_(initial), minalign_(1) {}

    // size of the finished data so far
    size_t size() const { return buf_.size() - head_; }
    const uint8_t *data() const { return buf_.data() + head_; }

    void align(size_t a) {
        if (a > minalign_) minalign_ = a;
    }
    // make room so that after writing `additional` bytes the write position is `align`-aligned
    void prep(size_t alignment, size_t additional) {
        align(alignment);
        size_t pad = (~(size() + additional) + 1) & (alignment - 1);
        grow(pad + alignment + additional);
        for (size_t i = 0; i < pad; i++) buf_[--head_] = 0;
    }
    template <class T> void push(T v) {
        prep(sizeof(T), 0);
        grow(sizeof(T));
        head_ -= sizeof(T);
        std::memcpy(&buf_[head_], &v, sizeof(T));
    }
    void push_bytes(const void *p, size_t n) {
        grow(n);
        head_ -= n;
        if (n) std::memcpy(&buf_[head_], p, n);
    }
    // offset (from the END of the buffer) of the current head = handle for later references
    uint32_t here() const { return (uint32_t)size(); }

    // push a uoffset32 referring to an object created earlier (handle = its `here()` value)
    void push_uoffset(uint32_t target) {
        prep(4, 0);
        uint32_t off = (uint32_t)size() + 4 - target;
        push<uint32_t>(off);
    }

    uint32_t create_string(const std::string &s) {
        prep(4, s.size() + 1);
        uint8_t z = 0;
        push_bytes(&z, 1);
        push_bytes(s.data(), s.size());
        push<uint32_t>((uint32_t)s.size());
        return here();
    }
    // vector of uoffsets to previously created objects
    uint32_t create_offset_vector(const std::vector<uint32_t> &targets) {
        prep(4, 4 * targets.size());
        for (size_t i = targets.size(); i-- > 0;) push_uoffset(targets[i]);
        push<uint32_t>((uint32_t)targets.size());
        return here();
    }
    // vector of fixed-size structs (elem_size bytes each, elem_align alignment), raw little-endian image
    uint32_t create_struct_vector(const void *elems, size_t count, size_t elem_size, size_t elem_align) {
        prep(4, elem_size * count);
        prep(elem_align, elem_size * count);
        push_bytes(elems, elem_size * count);
        push<uint32_t>((uint32_t)count);
        return here();
    }

    // ---- tables ----
    void start_table(int nfields) {
        fields_.assign(nfields, 0);
        table_start_ = here();
    }
    template <class T> void add_scalar(int field, T v, T def) {
        if (v == def) return;
        push<T>(v);
        fields_[field] = here();
    }
    template <class T> void add_scalar_force(int field, T v) {
        push<T>(v);
        fields_[field] = here();
    }
    void add_offset(int field, uint32_t target) {
        if (!target) return;
        push_uoffset(target);
        fields_[field] = here();
    }
    uint32_t end_table() {
        prep(4, 0);
        push<int32_t>(0);  // placeholder for the soffset to the vtable
        const uint32_t table = here();
        // trim trailing absent fields
        int n = (int)fields_.size();
        while (n > 0 && fields_[n - 1] == 0) n--;
        const uint16_t vt_size = (uint16_t)(4 + 2 * n);
        const uint16_t tbl_size = (uint16_t)(table - table_start_);
        // vtable goes in front of (below) the table
        prep(2, vt_size);
        for (int i = n; i-- > 0;) push<uint16_t>(fields_[i] ? (uint16_t)(table - fields_[i]) : 0);
        push<uint16_t>(tbl_size);
        push<uint16_t>(vt_size);
        const uint32_t vt = here();
        // patch the soffset: table_pos - vtable_pos (positions measured from buffer start) = vt - table in `here` units
        int32_t so = (int32_t)vt - (int32_t)table;
        std::memcpy(&buf_[buf_.size() - table], &so, 4);
        return table;
    }
    // root uoffset, buffer aligned to minalign
    void finish(uint32_t root) {
        prep(minalign_ > 4 ? minalign_ : 4, 4);
        push_uoffset(root);
    }

  private:
    void grow(size_t need) {
        if (head_ >= need) return;
        size_t old = buf_.size(), used = size();
        size_t ns = old * 2;
        while (ns - used < need) ns *= 2;
        std::vector<uint8_t> nb(ns);
        std::memcpy(nb.data() + ns - used, buf_.data() + head_, used);
        buf_.swap(nb);
        head_ = ns - used;
    }
    std::vector<uint8_t> buf_;
    size_t head_, minalign_;
    std::vector<uint32_t> fields_;
    uint32_t table_start_ = 0;
};

// ---- reading ----
// The buffer is untrusted (it comes from a file): every dereference is checked against [begin, end).  A field
// that cannot be read safely reads as absent (default value / empty string / empty vector / !ok() table).
struct Table {
    const uint8_t *base = nullptr;   // table position
    const uint8_t *begin = nullptr;  // the flatbuffer's bounds
    const uint8_t *end = nullptr;
    bool ok() const { return base != nullptr; }
    bool in(const uint8_t *p, uint64_t n) const { return p && p >= begin && p <= end && (uint64_t)(end - p) >= n; }
    uint16_t field_off(int field) const {
        if (!in(base, 4)) return 0;
        int32_t so;
        std::memcpy(&so, base, 4);
        const int64_t vt_pos = (int64_t)(base - begin) - (int64_t)so;
        if (vt_pos < 0 || vt_pos + 4 > (int64_t)(end - begin)) return 0;
        const uint8_t *vt = begin + vt_pos;
        uint16_t vt_size;
        std::memcpy(&vt_size, vt, 2);
        if (vt_size < 4 || !in(vt, vt_size)) return 0;
        const uint32_t idx = 4u + 2u * (uint32_t)field;
        if (idx + 2 > vt_size) return 0;
        uint16_t o;
        std::memcpy(&o, vt + idx, 2);
        return o;
    }
    template <class T> T scalar(int field, T def) const {
        const uint16_t o = field_off(field);
        if (!o || !in(base + o, sizeof(T))) return def;
        T v;
        std::memcpy(&v, base + o, sizeof(T));
        return v;
    }
    // target of an offset field: at least 4 readable bytes there (a length prefix or a table's soffset)
    const uint8_t *indirect(int field) const {
        const uint16_t o = field_off(field);
        if (!o || !in(base + o, 4)) return nullptr;
        uint32_t u;
        std::memcpy(&u, base + o, 4);
        if ((uint64_t)(end - (base + o)) < (uint64_t)u + 4) return nullptr;
        return base + o + u;
    }
    Table table(int field) const {
        Table t;
        t.base = indirect(field);
        t.begin = begin;
        t.end = end;
        return t;
    }
    std::string str(int field) const {
        const uint8_t *p = indirect(field);
        if (!p) return std::string();
        uint32_t n;
        std::memcpy(&n, p, 4);
        if (!in(p + 4, n)) return std::string();
        return std::string((const char *)p + 4, n);
    }
    // vector of `elem`-byte elements: element pointer + count (0 / nullptr if it does not fit the buffer)
    const uint8_t *vec(int field, uint32_t *count, uint32_t elem) const {
        *count = 0;
        const uint8_t *p = indirect(field);
        if (!p) return nullptr;
        uint32_t n;
        std::memcpy(&n, p, 4);
        if (!in(p + 4, (uint64_t)n * elem)) return nullptr;
        *count = n;
        return p + 4;
    }
    Table vec_table(int field, uint32_t i) const {
        uint32_t n;
        const uint8_t *e = vec(field, &n, 4);
        Table t;
        if (!e || i >= n) return t;
        uint32_t u;
        std::memcpy(&u, e + 4 * i, 4);
        const uint8_t *at = e + 4 * i;
        if ((uint64_t)(end - at) < (uint64_t)u + 4) return t;
        t.base = at + u;
        t.begin = begin;
        t.end = end;
        return t;
    }
};

inline Table root(const uint8_t *buf, size_t n) {
    Table t;
    if (n < 8) return t;
    uint32_t u;
    std::memcpy(&u, buf, 4);
    if ((uint64_t)u + 4 > n) return t;
    t.base = buf + u;
    t.begin = buf;
    t.end = buf + n;
    return t;
}

}  // namespace fb
