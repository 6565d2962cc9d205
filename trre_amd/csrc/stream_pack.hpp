// stream_pack.hpp — rows of transducer cells -> StreamTables (8-byte entries, pooled texts, the window
// form and the 16-byte count/emit form).  Shared by the two builders of stream-shaped tables:
// stream_build.cpp (the scan loop folded over raw bytes) and guided_build.cpp (the forward pass of the
// guided tables, whose columns are backward-pass symbols instead of byte classes).
#pragma once
#include <string>
#include <vector>

#include "front.hpp"

namespace trre {

struct StreamGiveUp {};          // the tables do not fit the formats: the caller falls back

struct StreamCell {
    uint32_t next = 0;
    std::string out;     // bytes emitted before the optional copy of the input byte
    bool copy_c = false;
    bool eol = false;    // this column ends the record
    bool ovf = false;    // bounded fold: the attempt outgrew the table (target SKIP, result void)
    bool diverge = false;   // guided tables: the reference's search would not terminate here
};

enum : uint8_t { kColPlain = 0, kColNewline = 1, kColNul = 2 };

struct StreamPackInput {
    std::vector<std::vector<StreamCell>> rows;   // [state][column]
    std::vector<uint8_t> col_kind;               // per column
    std::vector<uint32_t> pending_len;           // per state: bytes consumed but not yet emitted
    uint32_t skip = 1, done = 2;
    bool bounded = false;
    bool never_lp = false;                       // the caller knows the program is not length-preserving
    bool wide_cols = false;                      // more than 256 columns (16-bit symbols of a wide backward DFA): 8-byte entries only
};

// fills everything but StreamTables::cls
StreamTables pack_stream_tables(const StreamPackInput& in);

}  // namespace trre
