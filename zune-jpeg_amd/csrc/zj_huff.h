// zj_huff.h -- the GPU entropy stage for baseline Huffman scans (device side): the scan descriptor the
// CPU front-end prepares (zj_jpeg.cpp, prepare_scan) and the device code consumes (zj_huff_device.h, zj_huff.hip).
//
// What it replaces: the serial walk of src/mcu.rs:231-351 with src/bitstream.rs:314-373 (decode_mcu_block) per block.
// A Huffman stream has no index, so the scan is cut into SUB-SEQUENCES of at most 128 bytes, one per GPU thread, and
// decoded by self-synchronisation (Klein & Wiseman 2003; Weissenberger & Schmidt 2018/2021 for JPEG on GPUs):
//
//   round 0      every thread decodes its sub-sequence as if a block began at its first bit, and records where -- and in
//                which state (block of the MCU, zig-zag index) -- it crossed into the next sub-sequence;
//   round r      a sub-sequence whose predecessor's exit state changed in round r-1 is decoded again from that state (a
//                thread of round r-1 that changes its exit state puts its successor on round r's work list); a wrong
//                guess usually falls into step with the true parse within a few hundred bits, so the list
//                shrinks quickly; the rounds end when no exit state changed.  At that fixed point
//                exit[i] = F_i(exit[i-1]) for every i, and exit[first of a segment] started from the known state, so the
//                chain IS the serial parse;
//   scan         exclusive prefix sums over the blocks each sub-sequence completed and the DC differences it saw give
//                every thread its first block index and its DC predictors (segmented by restart interval);
//   write        every thread decodes once more and scatters the coefficients into the whole-frame planes the pixel
//                kernel reads (src/mcu_prog.rs:62-79 layout; round 0 has cleared them on the side);
//   cut          the MCUs the reference never decodes because its reader has come across EOI (zj_jpeg.cpp EoiCut) are cleared.
// Between the rounds a pass copies exit states forward through periodic runs (flat areas, below), where self-synchronisation
// fails.  Every kernel takes the working sets of up to HUFF_BATCH_MAX scans (blockIdx.y): several files, one launch per phase.
//
// Only the host can do the byte-level work cheaply ahead of time: it removes the stuffed zeros (T.81 B.1.1.5), cuts the
// scan at its RSTn markers into segments (each starts a fresh bit stream with zero predictors, T.81 E.1.4) and lays the
// sub-sequence grid over every segment.  All of it travels as ONE blob (header, tables, grid, bytes) = one H2D copy.
#pragma once

#include <stdint.h>

namespace zj {

constexpr int HUFF_SUB_MAX = 128;   // bytes a sub-sequence spans at most (its start is 16-byte aligned)
constexpr int HUFF_WG = 256;        // sub-sequences (threads) per workgroup: at most 32 KB of stream staged in LDS
constexpr int HUFF_L1_DC = 9;       // first-level window of a DC table (12 categories: longer codes are rare)
constexpr int HUFF_L1_AC = 11;      // ... of an AC table: a lane that meets a longer code sends its whole wave through
                                    // the second lookup, so "longer" has to be rare per WAVE (64 symbols), not per symbol
constexpr int HUFF_TAB_BUDGET = 7424; // u16 entries all tables of a scan may take together (14.5 KB of LDS: with the 37 KB
                                      // of stream words three workgroups fit a CU; two AC + two DC tables need 5120): a table is
                                      // 2^L1 + 2^(16 - L1) x (distinct L1-bit prefixes of its codes longer than L1 bits)
constexpr int HUFF_MAX_TABS = 6;    // distinct (class, id) tables of a 3-component scan
constexpr int HUFF_MAX_BPM = 10;    // blocks per MCU (T.81 B.2.3)
constexpr int HUFF_MAX_ROUNDS = 384; // synchronisation rounds before the scan is handed back to the CPU walker
constexpr uint32_t HUFF_MAGIC = 0x5a4a4853u;
constexpr uint32_t HUFF_FIRST = 0x80000000u, HUFF_LAST = 0x40000000u, HUFF_SEG_MASK = 0x3fffffffu;

// status bits the device raises; any of them sends the file to the CPU walker (zj_jpeg.cpp), which owns the
// reference-compatible treatment of damaged streams
constexpr uint32_t HUFF_ST_BAD_CODE = 1, HUFF_ST_RUN_OVER = 2, HUFF_ST_EXHAUSTED = 4, HUFF_ST_CUT_EARLY = 8,
                   HUFF_ST_PHASE = 16, HUFF_ST_NO_SYNC = 32,
                   // a DC symbol of more than 16 bits: the reference may read it short (src/bitstream.rs:278; zj_jpeg.cpp
                   // ref_dc_misread) depending on its reader's state, which only the CPU walker follows
                   HUFF_ST_DC_LONG = 64,
                   // a restart interval whose blocks are complete with a byte or more of data left in front of its marker, or
                   // whose marker the reference's reader has not come across by then (a last symbol of 26 bits or more): the
                   // reference resets nothing and decodes the NEXT interval out of what it holds, predictors and all -- only the
                   // CPU walker's serial walk reproduces that (round 6; tools/stream_soak.py, tools/ref_walk_soak.py)
                   HUFF_ST_LEFT_OVER = 128;

// table entry (u16): 16 (16 bits consumed, zig-zag advance 0) = no such code.  Bit 15 set: low byte = second-level table number, indexed by the 16 - L1 bits
// that follow the first L1.  Bit 15 clear: everything the symbol does to the parse --
//   bits 0-4   bits consumed: code length + magnitude bits (1..31)
//   bits 5-10  advance of the zig-zag index: DC symbols 1; AC run/size run + 1; ZRL 16; EOB (any run with size 0) 63
//   bits 11-14 magnitude bits (DC: the symbol itself; AC: its low nibble)
// (a DC symbol whose code + magnitude exceed 31 bits -- size 16, or 15 after a 16-bit code -- has no entry: the CPU walker
// accepts sizes up to 16 and gets such a file)
struct HuffBlk { uint8_t comp, hx, vy, pad; };                    // one block of the MCU, in scan order
struct HuffComp { uint32_t h, v, bw, bh; };                       // sampling factors, plane size in blocks
struct HuffSub { uint32_t start, seg; };                          // byte offset in the stream; segment | HUFF_FIRST | HUFF_LAST
struct HuffSeg { uint32_t start, end; };                          // byte range of a restart segment in the stream (end exact)

struct HuffScan { // header of the blob; every off_* is a byte offset from the header, 16-byte aligned
    uint32_t magic, blob_bytes;
    uint32_t nsub, nseg;
    uint32_t ri_mcus;          // MCUs per segment (the whole scan when there is no DRI)
    uint32_t bpm, ncomp, mcu_x, mcu_y, total_mcus;
    uint32_t is_eoi;           // the scan ends with EOI: the reference's early exit applies (zj_jpeg.cpp EoiCut)
    uint32_t rowlen;           // MCUs per row loop of the reference (mcu.rs:145-152)
    uint32_t tab_entries;      // u16 entries of all decoding tables together (even)
    uint32_t off_tab, off_sub, off_seg, off_stream, stream_bytes; // stream_bytes: multiple of 16, >= 32 zero bytes at the end
    uint32_t sub_bytes;        // nominal sub-sequence size (the last one of a segment is shorter)
    uint32_t round_budget;     // synchronisation rounds worth spending before the CPU walker is the faster way out
    uint32_t off_per, nper;    // periodic runs (below): per-sub-sequence words, how many are non-zero
    uint32_t comp_of_blk;      // 2 bits per block of the MCU: its component
    uint16_t dc_off[4], ac_off[4]; // per component: entry offset of its decoding tables
    HuffBlk blk[HUFF_MAX_BPM];
    HuffComp comp[3];
};

// Periodic runs.  A flat area is a run of identical tiny blocks: a periodic bit stream in which a decoder that is out of
// step settles into a cycle of its own and never meets the true parse, so the true state would advance one sub-sequence
// per round.  But where sub-sequence i holds the same bytes as sub-sequence i - q (q <= 8: the period in bits need not
// divide the sub-sequence size), the same entry state relative to its start gives the same exit state: once the second
// period of a run closes on itself (exit[a+q-1] = exit[a-1] shifted by q sub-sequences, a = second period), every later
// sub-sequence of the run is a shifted copy of its counterpart in that period.  The host finds the runs (memcmp against
// the previous eight sub-sequences); the periodic pass (huff_periodic_thread) applies the rule between rounds and puts
// what it changed on the next round's work list, where it is verified by decoding like everything else.
//   word i: 0, or (q << 28) | r0 for a sub-sequence the rule may predict: r0 = first sub-sequence of the run, i >= r0 + 2q
// the periodic pass runs in front of every second round (a run's second period has to be right before the rule can fire:
// ~2q rounds after the state in front of the run is; the pass itself is a few microseconds)
constexpr bool huff_periodic_before(int round) { return round >= 2 && (round & 1) == 0; }
constexpr uint32_t HUFF_LIST_FACTOR = 3; // a work list holds up to 3 x nsub entries
constexpr uint32_t HUFF_PER_QSHIFT = 28, HUFF_PER_MASK = (1u << 28) - 1, HUFF_PER_MAXQ = 8;

struct alignas(16) HuffI4 { int32_t x, y, z, w; };
// prefix sums (blocks completed, DC difference sums per component): a running value; `reset`: it is absolute because a
// restart segment began inside the range it covers
struct HuffAgg { int32_t v[4]; int32_t reset; };
constexpr int HUFF_SCAN_WG = 1024;  // sub-sequences per workgroup of the prefix-sum kernel
// control words
constexpr int HUFF_CTL_STATUS = 0, HUFF_CTL_SEEN = 1, HUFF_CTL_TICKET = 2, HUFF_CTL_ROUND0 = 3,
              HUFF_CTL_WORDS = HUFF_CTL_ROUND0 + HUFF_MAX_ROUNDS + 1;

// device-side working set of one scan
struct HuffArgs {
    const uint8_t* blob;       // device copy of the blob
    unsigned long long* exit;  // [nsub] packed exit state of every sub-sequence
    const unsigned long long* exit_rd; // where a round reads its predecessors' exit states: the same array (the emulation
                               // points it at a copy taken before the round -- on the device all threads of a round run at once)
    HuffI4* aux;               // [nsub] blocks completed, DC difference sums per component
    HuffI4* base;              // [nsub] first block index, DC predictors (after the scan kernel)
    uint32_t* list;            // [2][HUFF_LIST_FACTOR * nsub] from round 2 on: the sub-sequences a round has to decode again (those whose
                               // predecessor's exit state changed in the round before), compact, so that the waves of a
                               // sparse round are full; parity = round & 1, length = the change counter of the round before
    uint8_t* rel;              // [nsub] base[i] is relative to its prefix-sum workgroup (add wgpre)
    HuffAgg* wgagg;            // [ceil(nsub / HUFF_SCAN_WG)] totals of the prefix-sum workgroups
    HuffAgg* wgpre;            // the same, exclusive prefix
    uint32_t* ctl;             // HUFF_CTL_*: status bits, complement of the first MCU at which the reference has seen EOI, ticket of the
                               // prefix-sum workgroups, changes of round r at [HUFF_CTL_ROUND0 + r]
    int16_t* plane[3];
    uint8_t* zero_base;        // round 0 also clears the planes (the write pass stores non-zero coefficients only): the
    uint32_t zero_pieces;      // stores ride along with a latency-bound parse instead of a 50 MB fill of their own; 16-byte pieces
    int round;
    int spread;                // from round 2 on: work-list entry e is decoded by thread e * S, S = min(spread, nsub / entries):
                               // a wave with one active lane runs a symbol in 125 ns, a full one in 350 (every lane drags
                               // the others through its side paths) -- alone on the GPU a scan spreads out, in a batch it packs
};

// the working sets of the scans of one launch, passed BY VALUE in the kernel arguments (blockIdx.y picks one): read
// through a pointer instead, the same kernels ran the first full round 2.3x slower (152 vs 63 us)
constexpr int HUFF_BATCH_MAX = 16;
struct HuffBatch { HuffArgs a[HUFF_BATCH_MAX]; };

} // namespace zj
