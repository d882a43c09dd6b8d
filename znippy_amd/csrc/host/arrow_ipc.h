// Arrow IPC *stream* writer/reader for the two schemas of the Znippy container
// (znippy-common/src/index.rs:L43-54 sub-index, L279-288 manifest), written against the Arrow
// columnar/IPC specification (Message.fbs / Schema.fbs field order, encapsulated message framing).
// Layout choices follow arrow-rs 58.3's StreamWriter defaults as far as they are specified:
// metadata V5, 0xFFFFFFFF continuation marker, 64-byte alignment of message bodies and buffers,
// validity buffers omitted (length 0) for non-null columns, schema metadata keys sorted.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace aipc {

enum class Kind { Utf8, UInt32, UInt64, Bool, FixedBin32, Int8 };

struct Column {
    std::string name;
    Kind kind;
    bool nullable = false;
    // storage (only the member matching `kind` is used)
    std::vector<std::string> str;
    std::vector<uint32_t> u32;
    std::vector<uint64_t> u64;
    std::vector<uint8_t> u8;  // Bool: one byte per value; FixedBin32: 32 bytes per value; Int8: raw
    size_t rows() const;
};

struct Batch {
    std::vector<Column> cols;
    size_t rows() const { return cols.empty() ? 0 : cols[0].rows(); }
};

// One Arrow IPC stream: schema message, one record-batch message per batch, end-of-stream marker.
std::vector<uint8_t> write_stream(const std::vector<Batch> &batches, const Batch &schema_of,
                                  const std::map<std::string, std::string> &metadata);

// Parses a stream; appends every batch's rows to `out` (columns matched by schema order).
bool read_stream(const uint8_t *p, size_t n, Batch *out, std::map<std::string, std::string> *metadata,
                 std::string *err);

}  // namespace aipc
