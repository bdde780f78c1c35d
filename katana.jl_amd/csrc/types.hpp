// types.hpp -- what the HOST side of the engine needs to know about the kernels' data: launch constants and the few plain structs
// that sit by value in struct Engine.  Everything else (NlpDev, SweepOut, LpRows, SpMat, the packed records, ...) is defined next to
// the kernels that use it (kernels.hpp, mid_lp.hpp, batch_ecp.hpp) and only named here, so that a translation unit that launches no
// kernel -- the C ABI, abi.hip -- does not depend on the kernel headers.
#pragma once
#include <cstdint>

namespace ktn {

constexpr int kBlock = 256;
constexpr int kRedBlocks = 256;   // blocks of the two-stage deterministic reductions
constexpr int kChkQ = 16;         // quantities per check partial
constexpr int kBlkCols = 8192;    // columns of x* staged in LDS per workgroup of the column-blocked sweep (64 KB)

// peer-buffer transport (kernels.hpp "peer-buffer transport")
constexpr int kIpcMaxRanks = 8;
struct IpcPeers {
    double* data[kIpcMaxRanks];                    // rank r's exposed buffer: two slots of `cap` doubles
    unsigned long long* flags[kIpcMaxRanks];       // rank r's flag words, one per source rank
};

// tiled copy of a sparse matrix (kernels.hpp "tiled SpMV")
struct TiledMat {
    const int64_t* segstart;   // [tiles * nb_in + 1] first entry of (tile, block)
    const uint16_t* bptr;      // [tiles * nb_in * (kTileOut + 1)] entry offsets of the tile's outputs inside (tile, block)
    const uint16_t* idx;       // local input index
    const double* val;         // scaled values
    int nb_in;                 // input blocks
};

struct NlpDev; struct SweepOut; struct LpRows; struct SpMat;        // kernels.hpp
struct SepSlot; struct SepPartial; struct ColRec; struct RowRec;    // kernels.hpp
struct MidState;                                                     // mid_lp.hpp
struct EcpArena;                                                     // batch_ecp.hpp

}  // namespace ktn
