// fused_layout.h -- where the fused four-step kernel (rpf_fourstep.hip) keeps a tile's raw rows in LDS.  Plain C++ on
// integers, shared with the emulator under tests/ (tests/emul/fft_emul.cpp checks that the writer's and the reader's
// maps agree), like fft_core.h.
//
// A tile's rows are ROWB = 32, 64 or 128 bytes (16, 32 or 64 columns of u8 IQ pairs) and arrive in 16-byte pieces, 64
// pieces = 1 KB per LDS-DMA instruction, LDS piece q <- lane q % 64 of instruction q / 64.  Which (row, piece of the row)
// a lane fetches is free; where it lands is not (the instruction writes its 64 pieces back to back).
//   row-major   : piece q = PPR row + piece.  Consecutive rows are ROWB bytes apart: the 32 lanes of a ds_read_b32 group,
//                 which read one dword of 32 consecutive rows, meet in 4, 2 or 1 banks (8-, 16-, 32-way).
//   piece-major : inside each instruction's block of RPB = 64 / PPR rows, piece q % 64 = RPB piece + row % RPB.  The same
//                 piece of consecutive rows is 16 bytes apart: 8 banks, 4-way -- at the price of 16-byte memory requests
//                 that no longer join into a row.  Measured (profiles/r04_c4_fused.txt 6.): + 4 - 5 % at 64 and 128 bytes
//                 per row, - 0.5 % at 32, which stays row-major.
#pragma once

#include "fft_core.h"

namespace rpf {

template <int ROWB>
struct RawStage {
    static_assert(ROWB == 32 || ROWB == 64 || ROWB == 128, "16, 32 or 64 columns per tile");
    static constexpr int PPR = ROWB / 16;          // pieces per row
    static constexpr int RPB = 64 / PPR;           // rows per 1 KB block (one LDS-DMA instruction)
    static constexpr bool PIECE_MAJOR = ROWB > 32;
    // the row and the piece of that row that LDS piece q holds
    static RPF_HD int row_of(int q) { return PIECE_MAJOR ? RPB * (q / 64) + (q % 64) % RPB : q / PPR; }
    static RPF_HD int piece_of(int q) { return PIECE_MAJOR ? (q % 64) / RPB : q % PPR; }
    // LDS byte offset of byte `byte` (< ROWB) of row `row`
    static RPF_HD int offset(int row, int byte)
    {
        return PIECE_MAJOR ? 1024 * (row / RPB) + 16 * ((byte / 16) * RPB + row % RPB) + byte % 16 : row * ROWB + byte;
    }
};

}  // namespace rpf
