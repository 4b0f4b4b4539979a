// brotli_path_engine.h -- the path engine: the command engine of round 3 (included by brotli_kernels.hip inside its
// namespace, behind brotli_scan_engine.h, whose per-lane parsers it shares).
//
// What it replaces: the same serial walk as the scan engine (src/decode.rs:2359-2726, ProcessCommandsInternal, for
// metablocks whose literals do not depend on context), for the same blocks of sixteen waves that own one stream.  The scan
// engine parses a command at EVERY bit and lifts a literal-skip table over every bit; 92 % of the bits of the metric's
// streams are literal code words, and a command starts at one bit in a hundred.  The path engine does the work in proportion
// to what is there, region by region (PE_RBL stream bits, everything in LDS):
//
//   J1     the length of the literal code word that would start at every bit (decode.rs:378-398), eight bits per lane
//   path   the chain of literal code words from the region's first bit -- the LITERAL PATH: wherever the stream is inside a
//          literal run longer than a few symbols, its code words are the path's (prefix codes re-synchronise).  One lane
//          per 32 bits follows J1 from a guessed entry; the guesses are replaced by the exits of the chunks before until
//          nothing changes (what does not settle in PE_SYNC_ROUNDS rounds cuts the region short, nothing else).  Ranks by
//          prefix sum: por[rank] = bit, lit[rank] = the literal.  "n literals on from a bit of the path" is one lookup.
//   REC    a STATE is (bit, kind): kind E = "a distance code starts here, then a command" (decode.rs:2066-2131,
//          2134-2189), kind I = "a command starts here" (the command before had an implicit distance).  The record of a
//          state is the state its command's literal run ends in: head parsed, the run followed through J1 until it is on
//          the path (a few hops), the rest of it by rank.  Records are computed for every path bit as kind E (where literal
//          runs that met the path end), then for the states those records lead to that are not path states (runs that ended
//          before they met the path, implicit distances) -- round by round until nothing new turns up (the CLOSURE).  The
//          stream's own chain never leaves that set: its first state is seeded, and a record's successor is in the set by
//          construction.  ~27 records per real command instead of the scan engine's 96, and no lifting tables.
//   walk   wave 0 follows next[] from the region's entry eight commands a hop (NEXT8), the lanes behind an anchor fill in
//          theirs; a record that hit a cap (hops, closure room) is evaluated by the walker itself, uncapped.
//   details / resolve / execute
//          lane = command: each real command is parsed once more for its fields, then exactly the scan engine's resolve
//          (offsets, block counts, distance ring of decode.rs:2017-2049, every limit; the first command that needs the
//          checked loop ends the engine's part in front of it), then literals (out of lit[] by rank; the few before the run
//          met the path are decoded again) and LZ77 copies (decode.rs:2641-2680), the ones that read the region's own
//          output last and in order.
//
// Nothing depends on a guess: a record is the exact parse of its state or a marker (END: the parse leaves the region;
// BYHAND: a cap was hit), and the walk starts at the stream's real position.
// (no include guard: brotli_kernels.hip includes this file once per configuration -- PE_CFG_NS the namespace, PE_CFG_WAVES the
// waves of one engine, PE_CFG_RBL its region in stream bits, PE_CFG_PIPE whether two engines of a block take turns)
// PE_CFG_REMOTE (round 5): the engines that take a stream's regions in turns are BLOCKS -- a gang of up to eight, a CU each,
// on one stream (see "several CUs on one stream" below); what they tell each other goes through memory.)
#if !defined(PE_CFG_NS) || !defined(PE_CFG_WAVES) || !defined(PE_CFG_RBL) || !defined(PE_CFG_PIPE) || !defined(PE_CFG_DICT) || !defined(PE_CFG_REMOTE)
#error "brotli_path_engine.h: configuration macros missing"
#endif
#if PE_CFG_PIPE && PE_CFG_REMOTE
#error "brotli_path_engine.h: two engines a block or a gang of blocks, not both"
#endif
namespace PE_CFG_NS {
constexpr uint32_t GW = PE_CFG_WAVES;             // waves of one engine
constexpr bool PIPE2 = PE_CFG_PIPE != 0;          // two engines of GW waves a block, taking the stream's regions in turns
constexpr bool REMOTE = PE_CFG_REMOTE != 0;       // ... or one engine a block, and the blocks of a gang taking them in turns
constexpr bool PIPE = PIPE2 || REMOTE;            // (either way: a region's tables are built before the stream's entry into it is known)
static_assert(PIPE2 ? 2u * GW == SC_WAVES : GW == SC_WAVES, "engines and waves of a block");

constexpr uint32_t PE_RBL = PE_CFG_RBL;           // stream bits per region (local bit 0 = the first bit of the region's first dword)
constexpr uint32_t PE_CHUNKS = PE_RBL / 32;       // one lane per chunk of 32 bits: the whole block
constexpr uint32_t PE_RANKS = PE_RBL / 64u * 13u;  // path positions of a region at most (the region is cut where they run out): 6656 of 32 Kbit
constexpr uint32_t PE_WCAP = PE_RBL / 4u;         // closure states at most (records that would need more say BYHAND)
constexpr uint32_t PE_STATES = PE_RANKS + PE_WCAP;
constexpr uint32_t PE_GROW_BELOW = PE_WCAP / 4u;  // closure states below which a region that had been halved takes twice the bits again
constexpr uint32_t PE_HOPCAP = 8;                 // hops through J1 one evaluation takes; a run that needs more goes on in the lane's next evaluation
constexpr uint32_t PE_SYNC_ROUNDS = GW + 1;  // rounds between waves the chunk entries get to settle: enough for any code
constexpr uint32_t PE_CMDS = PE_RBL / 32u;          // commands one region's walk lists at most
#ifndef BROTLI_AMD_PE_DEPTH2_MIN
#define BROTLI_AMD_PE_DEPTH2_MIN 16
#endif
constexpr uint32_t PE_DEPTH2_MIN = BROTLI_AMD_PE_DEPTH2_MIN;   // a gang of this many blocks and more may have three executes under way (see the execute's waits)
constexpr uint32_t PE_DEP_ROUNDS = 6;             // levels of copies that build on each other which go side by side (execute); deeper ones in order
constexpr uint32_t PE_LANE_LITS = 64;             // literal runs up to this long are stored by their command's lane, four bytes a step
#ifndef BROTLI_AMD_PE_LANE_COPY
#define BROTLI_AMD_PE_LANE_COPY 16
#endif
constexpr uint32_t PE_LANE_COPY = BROTLI_AMD_PE_LANE_COPY;  // copies up to this long from in front of the region are done by their command's lane (16-byte loads, 16 .. 64)
static_assert(PE_LANE_COPY == 16, "lane copies: a region that is put together in LDS takes its short copies out of ONE sixteen-byte load (32 and 64 were round 3's experiment, before the stage)");
#ifndef BROTLI_AMD_PE_RUN_MIN
#define BROTLI_AMD_PE_RUN_MIN 6000
#endif
constexpr uint32_t PE_RUN_MIN = BROTLI_AMD_PE_RUN_MIN;             // literal runs from here on (about what a region's path holds) get regions of their own (2048 was tried: slower, C3 6.56 -> 6.75 ms:
                                                  // every run ends the invocation)
#ifndef BROTLI_AMD_PE_RUN_SB
#define BROTLI_AMD_PE_RUN_SB 256
#endif
constexpr uint32_t PE_RUN_SB = BROTLI_AMD_PE_RUN_SB;   // a long literal run's regions: stream bits a lane decodes one code word after the other (128 or 256: the longer the part,
                                                  // the likelier the lane's last word ends where it does whatever bit the lane entered at, and the fewer rounds the entries take)
constexpr uint32_t PE_RUN_RBL = 64u * GW * PE_RUN_SB;   // ... and the bits of such a region (no tables per bit: its input lies in the input's and J1's room)
constexpr uint32_t PE_MIN_INPUT = 4096;           // stream bits that must be left for a region to be worth its set-up
constexpr uint32_t PE_PIPE_MARGIN = 1024;          // two engines: a region's tables start this many bits in front of where the stream is expected to enter it
#ifndef BROTLI_AMD_PE_REMOTE_MARGIN
#define BROTLI_AMD_PE_REMOTE_MARGIN 1024
#endif
#ifndef BROTLI_AMD_PE_SEEDS
#define BROTLI_AMD_PE_SEEDS 384
#endif
#ifndef BROTLI_AMD_PE_SEED_BACK
#define BROTLI_AMD_PE_SEED_BACK 448
#endif
constexpr uint32_t PE_SEEDS = BROTLI_AMD_PE_SEEDS;          // a gang: entry seeds of a window built ahead of the stream (PEC_SEEDLO): this many bit positions ...
constexpr uint32_t PE_SEED_BACK = BROTLI_AMD_PE_SEED_BACK;  // ... from this far in front of where a stream that ran the region before to its end would enter
static_assert(!PE_CFG_REMOTE || PE_SEEDS <= 64u * PE_CFG_WAVES, "a thread a seed");
constexpr uint32_t PE_REMOTE_MARGIN = BROTLI_AMD_PE_REMOTE_MARGIN;   // a gang of blocks: the same margin between the windows of its plan
constexpr uint32_t PE_PIPE_USEFUL = 4096;          // ... and are used if the stream enters them with at least this many bits to go
constexpr uint32_t PE_PIPE_HAND = 48;              // ... and the walk evaluates this many states itself before the stream is on the path (commands without literals, one after the other)
constexpr uint32_t PE_PIPE_DECLINE = 2500;         // ... and a literal run from here on is the one-engine form's (its regions hold 6656 path positions, these half)
static_assert(PE_CHUNKS == 64u * GW, "one chunk per lane of the block");

// LDS layout, offsets from the engine's base (the scan engine's: the two never run at the same time)
constexpr uint32_t PE_CTL = 0;                                    // 1024: control words (the first 32 as the scan engine's), the stream's state
constexpr uint32_t PE_IN = 1024;                                   // the region's input: PE_RBL / 32 + 6 dwords
constexpr uint32_t PE_J1F = PE_IN + (PE_RBL / 32 + 8) * 4;        // code length at every bit, bit 7: on the path; later NEXT8
constexpr uint32_t PE_N8 = PE_J1F;                                // u16 per state: the state PE_JUMP commands on
#ifndef BROTLI_AMD_PE_JUMP_LOG
#define BROTLI_AMD_PE_JUMP_LOG 3
#endif
#ifndef BROTLI_AMD_PE_REMOTE_JUMP_LOG
#define BROTLI_AMD_PE_REMOTE_JUMP_LOG 4
#endif
// commands a hop of the walk (8 or 16; a gang of blocks: 16 -- there the walk is what the stream waits for, and one more doubling of the table is built ahead by somebody else)
constexpr uint32_t PE_JUMP_LOG = PE_CFG_REMOTE ? BROTLI_AMD_PE_REMOTE_JUMP_LOG : BROTLI_AMD_PE_JUMP_LOG, PE_JUMP = 1u << PE_JUMP_LOG;
static_assert(PE_JUMP_LOG == 3 || PE_JUMP_LOG == 4, "the walk's hop");
constexpr uint32_t PE_STG = PE_J1F;                               // the region's output while it is put together (execute), where it fits: see PE_STG_CAP
constexpr uint32_t PE_STG_CAP = PE_RBL;            // output bytes of a region that is put together in LDS and written out in one piece (0: never)
static_assert(PE_STG_CAP <= PE_RBL, "the stage lives in J1's room");
constexpr uint32_t PE_PM = PE_J1F + PE_RBL + 64;                  // u32 per chunk: which of its bits are on the path; later OFF
constexpr uint32_t PE_OFF = PE_PM;                                // u32 per listed command: where its output starts (from the region's)
constexpr uint32_t PE_CB = PE_PM + PE_CHUNKS * 4;                 // u16 per chunk: path positions in front of it
constexpr uint32_t PE_EX = PE_CB + PE_CHUNKS * 2 + 16;            // u8 per wave: where the chain leaves its last chunk (two buffers of 64)
constexpr uint32_t PE_POR = PE_EX + 2 * 64 + 32;                   // u16 per rank: its bit   (PE_EX: two buffers of a byte per wave)
constexpr uint32_t PE_LIT = PE_POR + PE_RANKS * 2;                // u8 per rank: its literal
constexpr uint32_t PE_NEXT = PE_LIT + PE_RANKS;                   // u16 per state: the state its command's literals end in
constexpr uint32_t PE_WST = PE_NEXT + PE_STATES * 2 + 16;         // u16 per closure state: bit | kind << 15; later the commands' records.  (The word in between, NEXT[PE_STATES], says PEN_NONE:
                                                                  // NEXT8's hops clamp their index to it instead of asking whether the state they stand on is one)
constexpr uint32_t PE_REC = PE_WST;                               // 16 bytes per listed command
constexpr uint32_t PE_BLIST = PE_NEXT + 2 * PE_CMDS;                // u16 per command with a long copy from in front of the region or a long literal run: its index
constexpr uint32_t PE_RS = PE_NEXT + 4 * PE_CMDS;                   // 64 bytes per batch of the resolve: its sums, the ring it ends with, its list counts
constexpr uint32_t PE_DLIST = PE_NEXT;                            // u16 per copy that reads the region's own output: its command (the records are dead by then)
constexpr uint32_t PE_WLIST = PE_NEXT + 4 * PE_CMDS + 1024;        // u16 per command whose copy is a word of the static dictionary that goes out inside the pass (PE_DICT)
static_assert(PE_WLIST + 2 * PE_CMDS <= PE_NEXT + PE_STATES * 2, "the words' list lies in the records' room");
static_assert(5u * PE_CMDS <= PE_RANKS * 2u, "the dependent copies' ranges and levels lie in the ranks' room");
constexpr uint32_t PE_WSTB = PE_WCAP * 2 > PE_CMDS * 16 ? PE_WCAP * 2 : PE_CMDS * 16;  // (bytes of that room: its larger tenant)
constexpr uint32_t PE_LIST = PE_WST + PE_WSTB;                    // u16 per listed command (+ 1): its state as bit | kind << 15
constexpr uint32_t PE_ANCH = PE_LIST + (PE_CMDS + 8) * 2;         // u32 per anchor of the walk: list index | state id << 16
constexpr uint32_t PE_TAILQ = PE_LIST;                              // u16 per state the bulk of the records left for the thin end (list and anchors are not in use then)
constexpr uint32_t PE_TAILCAP = PE_CMDS;
#ifndef BROTLI_AMD_PE_POLL_SLEEP
#define BROTLI_AMD_PE_POLL_SLEEP 2   // (x 64 clocks between two looks at what the walk has published)
#endif
#ifndef BROTLI_AMD_PE_TAIL_WAVES
#define BROTLI_AMD_PE_TAIL_WAVES 16   /* (all of an engine's, whatever their number) */
#endif
#ifndef BROTLI_AMD_PE_TAIL_AT
#define BROTLI_AMD_PE_TAIL_AT 80
#endif
constexpr uint32_t PE_TAIL_WAVES = BROTLI_AMD_PE_TAIL_WAVES;  // waves that see the thin end of the records through
constexpr uint32_t PE_TAIL_AT = BROTLI_AMD_PE_TAIL_AT;        // busy slots (of 128) below which a wave hands over what it holds
constexpr uint32_t PE_RUN_LIT = PE_POR;                            // a long literal run's region: its literals (the room of the ranks, literals, records and closure states)
constexpr uint32_t PE_RUN_WORK = 64u * GW * 2u * 3u;                 // (behind the literals: u16 per part -- its code words -- and two lists of parts whose entry has moved)
constexpr uint32_t PE_RUN_LITCAP = PE_LIST - PE_POR - 64u - PE_RUN_WORK;   // ... at most
constexpr uint32_t PE_RUN_CNT = PE_RUN_LIT + PE_RUN_LITCAP + 64u;  // u16 per part: code words that start in it
constexpr uint32_t PE_RUN_QA = PE_RUN_CNT + 64u * GW * 2u;         // u16 per entry, two lists in turns: part | (its new entry, bits into it) << 11
static_assert(PE_RUN_QA + 64u * GW * 4u <= PE_LIST && PE_RUN_CNT % 4 == 0 && 64u * GW <= 2048u, "a run region's work lists");
constexpr uint32_t PE_RUN_EX = PE_PM;                              // ... u8 per lane: where its last code word ends (bits into the next lane's part)
static_assert((PE_RUN_RBL / 32u + 8u) * 4u <= PE_PM - PE_IN && 64u * GW <= PE_CHUNKS * 4u, "a run region's input and exits");
static_assert(PE_RUN_LIT % 16 == 0 && PE_STG % 16 == 0, "what write_out reads line by line");
static_assert(PIPE || 64u * GW * 2u + 4096u <= PE_EX - PE_PM, "a run region's exits and its table of the literal code (one engine: two never take a run)");
constexpr uint32_t PE_SET_BYTES = PE_ANCH + 128 * 4;              // one engine's tables
// What the engines of a block share: the invocation's parameters, the stream's state, the records' two tables.  One engine: at the
// end of its tables (the control words are its own); two engines: in front of theirs, with a block of control words of its own.
// A record is parsed where 128 bits of the region are left in front of it (pos + 128 <= L); its reads reach further: a distance
// code of up to 15 + 62 bits (large window) in front of a head whose 64-bit read takes three dwords, i.e. bit pos + 77 + 96 at
// most -- 45 bits beyond L, inside the six dwords (192 bits) of input that every region stages behind its last one.
static_assert(15u + 62u + 96u <= 128u + 6u * 32u, "a record's reads stay inside the region's input slack");
constexpr uint32_t PE_TD_ENTRIES = 1024;                          // (920 is the most a distance alphabet without large window takes; a large-window table that needs more keeps its metablock off the engine: td_ok)
constexpr uint32_t PE_SHARED_CTL = PIPE2 ? 1024u : 0u;            // the shared control words (two engines)
constexpr uint32_t PE_TD = PIPE2 ? PE_SHARED_CTL : PE_SET_BYTES;  // u16 per entry of the distance code's table: the same two levels, a leaf's value = bits of the whole distance code (symbol + extra)
constexpr uint32_t PE_TC = PE_TD + PE_TD_ENTRIES * 2;             // u32 per command symbol: insert base | insert extra bits << 15 | copy extra bits << 20 | implicit distance << 25
constexpr uint32_t PE_SET0 = PIPE2 ? PE_TC + 704 * 4 : 0u;        // the first engine's tables
constexpr uint32_t PE_BYTES = PIPE2 ? PE_SET0 + 2u * PE_SET_BYTES : PE_TC + 704 * 4;
static_assert(PE_BYTES <= SC_BYTES, "the path engine lives in the scan engine's LDS");
static_assert(PE_STATES * 2 <= PE_RBL + 64 && PE_WCAP * 2 <= PE_WSTB && PE_CMDS * 4 <= PE_CHUNKS * 4 && PE_STATES % 8 == 0, "overlays");
static_assert(PE_J1F % 16 == 0 && PE_PM % 16 == 0 && PE_REC % 16 == 0 && PE_POR % 4 == 0 && PE_NEXT % 4 == 0 && PE_LIST % 4 == 0 && PE_TD % 4 == 0 && PE_TC % 4 == 0 && PE_SET0 % 16 == 0 && PE_SET_BYTES % 16 == 0, "alignment");

enum { PEN_END = 0xFFFFu, PEN_BYHAND = 0xFFFEu, PEN_NONE = 0xFFFDu, PEN_FIRST_SPECIAL = 0xFFF0u };
// control words of a region (from 64 on; the invocation's parameters are the scan engine's SCC_*)
enum { PEC_LBDW = 64, PEC_LE = 65, PEC_L = 66, PEC_LP = 67, PEC_RN = 68, PEC_WN = 69, PEC_TMIN = 70, PEC_M = 71, PEC_GO = 72, PEC_KP = 73,
       PEC_P0_LO = 74, PEC_P0_HI = 75, PEC_ANYDEP = 76, PEC_CHG = 77 /* three words */, PEC_STATE = 160 /* the stream's state between wave 0's uses of it: PeStream */, PEC_CONT = 96, PEC_NEXT_LBDW = 97, PEC_ON = 98, PEC_NA = 99, PEC_NBIG = 101, PEC_TAILN = 102, PEC_TAILNEXT = 103, PEC_READY = 104, PEC_ENT = 105, PEC_MODE = 106, PEC_DEPCHG = 107 /* the dependent copies' levels: bit r, round r changed one */, PEC_DEPLV0 = 108 /* ... levels of the copies that do not lag */, PEC_DEPLV1 = 109 /* ... and of those that do */, PEC_DEPDEEP = 110 /* ... some are deeper than the rounds go */, PEC_PREVOUT = 111 /* (a gang) the bytes of the region before's output */, PEC_DEPTH = 112 /* ... how many regions before this one its resolve took for still under way: PEC_RELAX when the stream arrived */, PEC_TAKE = 107, PEC_BKP = 108 /* + batch: 16 words */, PEC_NEXTRANK = 100, PEC_WSUM = 80 /* + wave: 16 words */, PEC_NAPUB = 125 /* anchors the walk has published */, PEC_WDONE = 126 /* the walk is over */,
       PEC_STAGED = 127 /* the region's output is put together in LDS */, PEC_OUTTOT = 128 /* its size */, PEC_TDN = 129 /* entries of the distance code's table */, PEC_SCRATCH = 130 /* stores that are not meant land here */, PEC_GBAR = 131 /* the engine's barrier: arrivals so far */,
       // two engines (words of the shared block): what they tell each other
       PEC_RESOLVED = 132 /* regions whose resolve is through: the stream's state is the next one's */, PEC_EXECUTED = 133 /* regions whose output is in memory */,
       PEC_STOP = 134 /* the invocation is over */, PEC_NFINAL = 135 /* regions whose window is final */, PEC_WINF = 136 /* + (region & 1): its first dword */,
       PEC_DECLINE = 139 /* the next command's literal run wants regions of its own: the one-engine form's */, PEC_PLAN = 140 /* (an engine's own word) what to do with the tables it built */, PEC_MYENTRY = 141 /* ... where the stream entered its region */, PEC_MYNEXT = 142 /* ... and where it left it */,
       PEC_BIGNEXT = 143 /* the execute's items that get a wave: handed out so far */, PEC_NXOK = 144 /* the number of the region whose PEC_CONT / PEC_NEXT_LBDW are there */ , PEC_KS = 145 /* the pass's first command (passes: see PE_DICT) */, PEC_DICTK = 146 /* the command whose copy is a word of the static dictionary, its literals out: its index, distance, copy length */, PEC_DICTD = 147, PEC_DICTN = 148, PEC_AGAIN = 149, PEC_PDX = 150 /* the invocation ends behind that command's distance (SCX_POST_DISTANCE) */, PEC_OVF = 151 /* regions of this invocation whose closure all but filled its room */,
       PEC_MYGEN = 157 /* (a gang) the generation of the plan this engine's window follows */, PEC_MYSHIFT = 138 /* ... and how often its regions are halved */, PEC_MEMBERS = 186 /* ... (a pool) the blocks of this invocation's gang */, PEC_NOHELP = 185 /* ... (a pool) the owner kept the invocation to itself because nobody has joined its stream */, PEC_RELAX = 159 /* ... whether this engine's executes wait twice (see there): its own observation, kept from region to region */, PEC_LAG = 137 /* ... the bytes of the region before's output: what a copy may not read before that region's engine says they are there */, PEC_BUILT = 158 /* ... whether its tables are built */,
       PEC_SEEDLO = 190 /* (a gang) the window's ENTRY SEEDS: the states 'a command starts at bit PEC_SEEDLO + i', i < PEC_SEEDN, are closure states 1 + i -- evaluated with the
                            tables, ahead of the stream, so that the walk finds the state the stream enters in among them instead of evaluating it itself (see the walk) */, PEC_SEEDN = 191,
       PEC_FIN = 156 /* a long literal run has ended in this region: its command's distance and copy are wave 0's, in place */,
       PEC_DSEEN = 155 /* (lean form) the engine's part ended in front of a dictionary reference: the general form's stream */,
       PEC_DCAND = 152 /* (PE_DICT) a command of the pass may be a word of the static dictionary */, PEC_NWORD = 153 /* ... words the pass puts out */, PEC_WNEXT = 154 /* ... handed out so far */ };
// Words of the static dictionary (decode.rs:2593-2640; one command in 33 to 87 of text at -q 4 .. 9, tools/eligibility_survey.py).
// Round 4's engine stopped in front of each: an invocation and a region's tables for some fifty commands, 3200 clocks a command.
// Now (one engine): the resolve lets the first such command of what is listed through with its literals alone, wave 0 puts the
// word behind them when the pass's output is complete, and the resolve and the execute run again for the commands behind it --
// the region's tables, the walk's list and the details in the waves' registers hold for every pass.  A word that is not a plain
// one (unknown transform, an empty word, one that does not fit the limits) ends the invocation behind the command's distance:
// the checked loop says what it is.
// The engine compiles twice for that (round 5): PE_CFG_DICT 0 is the LEAN form -- it stops in front of a dictionary reference, as round 3's
// did, and tells the caller, who takes the general form (PE_CFG_DICT 1) for the rest of the stream.  A stream without such words -- the
// metric's -- never runs the general form: what that form carries had cost it 4 % through the allocation of one very large function's
// registers (128 a wave, and the function spills).
#if PE_CFG_DICT && !PE_CFG_PIPE && !PE_CFG_REMOTE && !defined(BROTLI_AMD_PE_NO_DICT)
#define PE_DICT 1
#else
#define PE_DICT 0
#endif


#ifdef BROTLI_AMD_PROFILE_SCAN
#ifndef BROTLI_AMD_PATH_PROF_DEFINED
#define BROTLI_AMD_PATH_PROF_DEFINED
}  // namespace
__device__ unsigned long long g_path_prof[40];
namespace PE_CFG_NS {
#endif
#define PE_PROF(k) do { if (me == 0) { uint64_t _t = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0) pp_acc[k] += _t - pp_t; pp_t = _t; } } while (0)
#define PE_COUNT(k, v) do { if (me == 0 && blockIdx.x == 0) pp_acc[k] += (v); } while (0)
#define PE_LANECOUNT(k, cond) do { if (blockIdx.x == 0 && (cond)) atomicAdd(&g_path_prof[k], 1ull); } while (0)
#else
#define PE_LANECOUNT(k, cond) do { } while (0)
#define PE_PROF(k) do { } while (0)
#define PE_COUNT(k, v) do { } while (0)
#endif

typedef __attribute__((address_space(3))) uint32_t pe_lds_u32;
__device__ __forceinline__ uint32_t pe_ctl_ld(uint32_t pb, uint32_t k) { return rfl(*reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * k])); }
__device__ __forceinline__ void pe_ctl_st(uint32_t pb, uint32_t k, uint32_t v) { if (lane_id() == 0) *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * k]) = v; }
__device__ __forceinline__ uint32_t pe_atomic_add(uint32_t addr, uint32_t v) {
  return __hip_atomic_fetch_add(reinterpret_cast<pe_lds_u32*>(&g_smem[addr]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ... by lane 0 alone, the old value in every lane's hands (uniform): the compiler's own form of an atomic inside `if (lane == 0)`
// counts the active lanes and multiplies first -- twenty instructions where these six do
__device__ __forceinline__ uint32_t pe_atomic_add_uniform(uint32_t addr, uint32_t v) {
  uint32_t r; uint64_t sv;
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %[r], %[a], %[v]\n\ts_mov_b64 exec, %[sv]\n\ts_waitcnt lgkmcnt(0)"
               : [r] "=&v"(r), [sv] "=&s"(sv) : [a] "v"(addr), [v] "v"(v) : "memory");
  return rfl(r);
}
__device__ __forceinline__ uint32_t pe_atomic_or(uint32_t addr, uint32_t v) {
  return __hip_atomic_fetch_or(reinterpret_cast<pe_lds_u32*>(&g_smem[addr]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t pe_atomic_max(uint32_t addr, uint32_t v) {
  return __hip_atomic_fetch_max(reinterpret_cast<pe_lds_u32*>(&g_smem[addr]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t pe_atomic_min(uint32_t addr, uint32_t v) {
  return __hip_atomic_fetch_min(reinterpret_cast<pe_lds_u32*>(&g_smem[addr]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the 64 stream bits from local bit p of the region
__device__ __forceinline__ void pe_bits64(uint32_t pb, uint32_t p, uint32_t& lo, uint32_t& hi) {
  const uint32_t q = pb + PE_IN + ((p >> 5) << 2), sh = p & 31u;
  const uint32_t a0 = lds_ld32(q), a1 = lds_ld32(q + 4u), a2 = lds_ld32(q + 8u);
  lo = __builtin_amdgcn_alignbit(a1, a0, sh);
  hi = __builtin_amdgcn_alignbit(a2, a1, sh);
}
__device__ __forceinline__ uint32_t pe_bits32(uint32_t pb, uint32_t p) {
  const uint32_t q = pb + PE_IN + ((p >> 5) << 2);
  return __builtin_amdgcn_alignbit(lds_ld32(q + 4u), lds_ld32(q), p & 31u);
}


// The stream's state (wave 0's, uniform).  It lives in LDS between the places that use it -- the region's set-up, the
// resolve, a literal run's regions, the hand-over --, so that the registers it would take do not spill in the loops between.
struct PeStream {
  uint64_t P; uint32_t quota, bl0, bl1, bl2, ncmd, b, rbl, run_on, run_rem, run_copy, run_implicit, run_dctx;
  int32_t mlen, d0, d1, d2, d3, max_backward;
  uint32_t first;  // the invocation's first region is still to come
  uint32_t s_bits, s_cmds, s_lits, s_dsts;  // what the region (or pass) before took: stream bits, commands, literals, distance codes -- the next region is as long as its block counts last at that rate
};
__device__ __forceinline__ PeStream pe_st_load(uint32_t pb) {
  PeStream st;
  // (one LDS read for all of it: lane k reads word k, the fields come out of the lanes)
  const uint32_t v = *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * (PEC_STATE + lane_id())]);
  st.P = (uint64_t)rdlane(v, 0) | ((uint64_t)rdlane(v, 1) << 32);
  st.quota = rdlane(v, 2); st.bl0 = rdlane(v, 3); st.bl1 = rdlane(v, 4); st.bl2 = rdlane(v, 5);
  st.ncmd = rdlane(v, 6); st.b = rdlane(v, 7); st.rbl = rdlane(v, 8); st.run_on = rdlane(v, 9);
  st.run_rem = rdlane(v, 10); st.run_copy = rdlane(v, 11); st.run_implicit = rdlane(v, 12); st.run_dctx = rdlane(v, 13);
  st.mlen = (int32_t)rdlane(v, 14); st.d0 = (int32_t)rdlane(v, 15); st.d1 = (int32_t)rdlane(v, 16);
  st.d2 = (int32_t)rdlane(v, 17); st.d3 = (int32_t)rdlane(v, 18); st.max_backward = (int32_t)rdlane(v, 19); st.first = rdlane(v, 20);
  st.s_bits = rdlane(v, 21); st.s_cmds = rdlane(v, 22); st.s_lits = rdlane(v, 23); st.s_dsts = rdlane(v, 24);
  return st;
}
__device__ __forceinline__ void pe_st_store(uint32_t pb, const PeStream& st) {
  const uint32_t a = PEC_STATE;
  pe_ctl_st(pb, a, (uint32_t)st.P); pe_ctl_st(pb, a + 1, (uint32_t)(st.P >> 32));
  pe_ctl_st(pb, a + 2, st.quota); pe_ctl_st(pb, a + 3, st.bl0); pe_ctl_st(pb, a + 4, st.bl1); pe_ctl_st(pb, a + 5, st.bl2);
  pe_ctl_st(pb, a + 6, st.ncmd); pe_ctl_st(pb, a + 7, st.b); pe_ctl_st(pb, a + 8, st.rbl); pe_ctl_st(pb, a + 9, st.run_on);
  pe_ctl_st(pb, a + 10, st.run_rem); pe_ctl_st(pb, a + 11, st.run_copy); pe_ctl_st(pb, a + 12, st.run_implicit); pe_ctl_st(pb, a + 13, st.run_dctx);
  pe_ctl_st(pb, a + 14, (uint32_t)st.mlen); pe_ctl_st(pb, a + 15, (uint32_t)st.d0); pe_ctl_st(pb, a + 16, (uint32_t)st.d1);
  pe_ctl_st(pb, a + 17, (uint32_t)st.d2); pe_ctl_st(pb, a + 18, (uint32_t)st.d3); pe_ctl_st(pb, a + 19, (uint32_t)st.max_backward); pe_ctl_st(pb, a + 20, st.first);
  pe_ctl_st(pb, a + 21, st.s_bits); pe_ctl_st(pb, a + 22, st.s_cmds); pe_ctl_st(pb, a + 23, st.s_lits); pe_ctl_st(pb, a + 24, st.s_dsts);
}

// What every phase needs to know about the region (uniform)
struct PeCtx {
  uint32_t pb, td, tc, lit_tree, cmd_tree, dtree, postfix_bits, num_direct, lut_vgpr;   // (pb: the engine's tables; td, tc: the records' tables)
  uint32_t L;    // bits of the region that may be parsed (a record needs 128 in front of it)
  uint32_t Lp;   // the path ends in front of this bit
  uint32_t Rn;   // path positions
};
// A state's record, with everything the later phases want from the same parse.
struct PeParse {
  uint32_t hy, hn;   // code 3: where the run stands (bit, literals left)
  uint32_t code;     // 0: next state is path state `next` (a rank); 1: next state is `next` = bit | kind << 15, not a path state; 2: END; 3: BYHAND
  uint32_t next;
  uint32_t p, x;     // the command's first bit, its literals' first bit
  uint32_t insert, copy, implicit;
  uint32_t u, ry;    // literals before the run is on the path, rank of the first one that is (meaningful when u < insert)
  uint32_t dkind, dval;  // kind E: the distance code parsed at the state's bit (the distance of the command BEFORE)
};
// rank of path bit y
__device__ __forceinline__ uint32_t pe_rank(uint32_t pb, uint32_t y) {
  return lds_ld16(pb + PE_CB + ((y >> 5) << 1)) + (uint32_t)__builtin_popcount(lds_ld32(pb + PE_PM + ((y >> 5) << 2)) & ((1u << (y & 31u)) - 1u));
}
// The records of NS states per lane, side by side: state t of the lane is (pos[t], kind[t]); `on[t]` says whether it is one.
// Every lane of the wave must call it (the parsers use cross-lane permutes).  The chains of dependent LDS reads of the NS
// states are written stage by stage, so that their round trips overlap (an evaluation is some twenty of them in a row).
// CAPPED: the hop limit of the table rounds; J1: hop through the J1 table (else: decode the literal code words again, for
// the phases that run when J1's room holds NEXT8).
// FULL: every field of the parse; otherwise only what the records need (where the run ends, no values).
// A capped evaluation that ran out of hops says code 3 and leaves where it stands in r.hy / r.hn / r.implicit: given back
// through `res`, the next call goes on from there (RESUME).
struct PeResume { bool on; uint32_t y, n, implicit; };
template <uint32_t NS, bool CAPPED, bool J1, bool RESUME = false, bool FULL = true>
__device__ __forceinline__ void pe_eval_n(const PeCtx& c, const uint32_t (&pos)[NS], const uint32_t (&kind)[NS], const bool (&on)[NS], PeParse (&r)[NS], const PeResume* res = nullptr) {
  const uint32_t pb = c.pb;
  bool ok[NS]; uint32_t q[NS], p[NS], lo[NS], hi[NS];
  bool any_e = false;
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    ok[t] = on[t] && pos[t] + 128u <= c.L; q[t] = ok[t] ? pos[t] : 0u; p[t] = q[t];
    r[t].dkind = SCK_IMPLICIT; r[t].dval = 0;
    any_e = any_e || (ok[t] && kind[t] == 0u);
  }
  if (__ballot(any_e) != 0ull) {
    // the distance code at the state's bit (sc_dist, the NS lookups side by side)
    uint32_t e[NS], Ld[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { if (FULL) pe_bits64(pb, q[t], lo[t], hi[t]); else { lo[t] = pe_bits32(pb, q[t]); hi[t] = 0u; } }
    SC_STAGE();
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) e[t] = lds_ld16(c.dtree + ((lo[t] & 0xFFu) << 1));
    SC_STAGE();
    bool sec = false;
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { Ld[t] = e[t] & 15u; sec = sec || Ld[t] > ROOT_BITS; }
    if (__ballot(sec) != 0ull) {
      uint32_t e2[NS];
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
        const bool s2 = Ld[t] > ROOT_BITS;
        const uint32_t idx = s2 ? (e[t] >> 4) + __builtin_amdgcn_ubfe(lo[t], ROOT_BITS, Ld[t] - ROOT_BITS) : (lo[t] & 0xFFu);
        e2[t] = lds_ld16(c.dtree + (idx << 1));
      }
      SC_STAGE();
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) if (Ld[t] > ROOT_BITS) { e[t] = e2[t]; Ld[t] = ROOT_BITS + (e2[t] & 15u); }
    }
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      if (FULL) {
        const ScDist d = sc_dist_finish(e[t] >> 4, Ld[t], lo[t], hi[t], c.postfix_bits, c.num_direct);
        if (kind[t] == 0u) { p[t] = q[t] + d.bits; r[t].dkind = d.kind; r[t].dval = d.val; }
      } else {  // only the code's length: the symbol's bits and its extra bits (ReadDistanceInternal, decode.rs:2099-2128)
        const uint32_t code = e[t] >> 4;
        const int32_t dv = (int32_t)code - (int32_t)c.num_direct;
        const uint32_t nb = (code >= 16u && dv >= 0) ? (((uint32_t)dv >> c.postfix_bits) >> 1) + 1u : 0u;
        if (kind[t] == 0u) p[t] = q[t] + Ld[t] + nb;
      }
    }
  }
  // the command's head (sc_head)
  uint32_t y[NS], n[NS], imp[NS];
  {
    uint32_t e[NS], Lh[NS], ie[NS], ce[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) pe_bits64(pb, p[t], lo[t], hi[t]);
    SC_STAGE();
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) e[t] = lds_ld16(c.cmd_tree + ((lo[t] & 0xFFu) << 1));
    SC_STAGE();
    bool sec = false;
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { Lh[t] = e[t] & 15u; sec = sec || Lh[t] > ROOT_BITS; }
    if (__ballot(sec) != 0ull) {
      uint32_t e2[NS];
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
        const bool s2 = Lh[t] > ROOT_BITS;
        const uint32_t idx = s2 ? (e[t] >> 4) + __builtin_amdgcn_ubfe(lo[t], ROOT_BITS, Lh[t] - ROOT_BITS) : (lo[t] & 0xFFu);
        e2[t] = lds_ld16(c.cmd_tree + (idx << 1));
      }
      SC_STAGE();
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) if (Lh[t] > ROOT_BITS) { e[t] = e2[t]; Lh[t] = ROOT_BITS + (e2[t] & 15u); }
    }
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      const uint32_t cmd = e[t] >> 4, cell = cmd >> 6;
      const uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((cmd >> 3) & 7u);
      const uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (cmd & 7u);
      ie[t] = bperm(ins_code << 2, c.lut_vgpr); ce[t] = bperm((32u + copy_code) << 2, c.lut_vgpr);
      imp[t] = cmd < 128u ? 1u : 0u;
    }
    SC_STAGE();
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      uint64_t w = (((uint64_t)hi[t] << 32) | lo[t]) >> Lh[t];
      const uint32_t ib = ie[t] >> 16, cb = ce[t] >> 16;
      r[t].insert = (ie[t] & 0xFFFFu) + ((uint32_t)w & ((1u << ib) - 1u));
      w >>= ib;
      r[t].copy = (ce[t] & 0xFFFFu) + ((uint32_t)w & ((1u << cb) - 1u));
      r[t].p = p[t]; r[t].x = p[t] + Lh[t] + ib + cb; r[t].implicit = imp[t];
      y[t] = r[t].x; n[t] = r[t].insert;
      if (RESUME && res[t].on) { y[t] = res[t].y; n[t] = res[t].n; imp[t] = res[t].implicit; r[t].implicit = imp[t]; }
    }
  }
  // the literal runs: hop by hop until they are on the path (or over), the rest by rank
  uint32_t hops[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) hops[t] = 0;
  if (CAPPED && J1) {
    // (the table rounds: the bytes of J1 from Lp on carry the path flag -- sentinels --, so a run that reaches them stops
    // there, and a lane without a state, or whose run starts beyond them, has nothing to hop)
    uint32_t m[NS]; bool part[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { part[t] = (bool)((uint32_t)ok[t] & (uint32_t)(y[t] < c.Lp)); m[t] = part[t] ? n[t] : 0u; }
#ifndef BROTLI_AMD_PE_NO_HOP_ASM
    if constexpr (NS <= 2u) {
      // The same loop by hand.  A state that has stopped hopping -- no literals left, or on the path -- stays stopped, so the
      // lanes still hopping are an execution mask that only ever narrows: v_cmpx drops the lanes, the step itself is an
      // add and a decrement (four vector instructions a hop and state; the compiled form below takes ten).
      const uint32_t jb = pb + PE_J1F;
      uint32_t ya0 = y[0] + jb, ya1 = y[NS - 1u] + jb, m0 = m[0], m1 = m[NS - 1u], f0, f1;
      uint64_t e0, e1, ea;
#define PE_HOP1 \
      "s_mov_b64 exec, %[e0]\n\tds_read_u8 %[f0], %[y0]\n\ts_waitcnt lgkmcnt(0)\n\tv_cmpx_gt_u32 vcc, %[c80], %[f0]\n\tv_add_u32 %[y0], %[y0], %[f0]\n\t" \
      "v_subrev_u32 %[m0], 1, %[m0]\n\tv_cmpx_ne_u32 vcc, 0, %[m0]\n\ts_mov_b64 %[e0], exec\n\ts_cmp_eq_u64 %[e0], 0\n\ts_cbranch_scc1 .Lpe_hop_done_%=\n\t"
#define PE_HOP2 \
      "s_mov_b64 exec, %[e0]\n\tds_read_u8 %[f0], %[y0]\n\ts_mov_b64 exec, %[e1]\n\tds_read_u8 %[f1], %[y1]\n\ts_mov_b64 exec, %[e0]\n\ts_waitcnt lgkmcnt(1)\n\t" \
      "v_cmpx_gt_u32 vcc, %[c80], %[f0]\n\tv_add_u32 %[y0], %[y0], %[f0]\n\tv_subrev_u32 %[m0], 1, %[m0]\n\tv_cmpx_ne_u32 vcc, 0, %[m0]\n\ts_mov_b64 %[e0], exec\n\t" \
      "s_mov_b64 exec, %[e1]\n\ts_waitcnt lgkmcnt(0)\n\t" \
      "v_cmpx_gt_u32 vcc, %[c80], %[f1]\n\tv_add_u32 %[y1], %[y1], %[f1]\n\tv_subrev_u32 %[m1], 1, %[m1]\n\tv_cmpx_ne_u32 vcc, 0, %[m1]\n\ts_mov_b64 %[e1], exec\n\t" \
      "s_or_b64 %[ea], %[e0], %[e1]\n\ts_cbranch_scc0 .Lpe_hop_done_%=\n\t"
      if constexpr (NS == 1u) {
        asm volatile("s_mov_b64 %[ea], exec\n\tv_cmp_ne_u32 %[e0], 0, %[m0]\n\ts_cmp_eq_u64 %[e0], 0\n\ts_cbranch_scc1 .Lpe_hop_done_%=\n\t"
                     PE_HOP1 PE_HOP1 PE_HOP1 PE_HOP1 PE_HOP1 PE_HOP1 PE_HOP1 PE_HOP1
                     ".Lpe_hop_done_%=:\n\ts_mov_b64 exec, %[ea]"
                     : [y0] "+v"(ya0), [m0] "+v"(m0), [f0] "=&v"(f0), [e0] "=&s"(e0), [ea] "=&s"(ea)
                     : [c80] "v"(0x80u) : "vcc", "scc", "memory");
        (void)ya1; (void)m1; (void)f1; (void)e1;
        y[0] = ya0 - jb; m[0] = m0;
      } else {
        uint64_t sv;
        asm volatile("s_mov_b64 %[sv], exec\n\tv_cmp_ne_u32 %[e0], 0, %[m0]\n\tv_cmp_ne_u32 %[e1], 0, %[m1]\n\ts_or_b64 %[ea], %[e0], %[e1]\n\ts_cbranch_scc0 .Lpe_hop_done_%=\n\t"
                     PE_HOP2 PE_HOP2 PE_HOP2 PE_HOP2 PE_HOP2 PE_HOP2 PE_HOP2 PE_HOP2
                     ".Lpe_hop_done_%=:\n\ts_mov_b64 exec, %[sv]"
                     : [y0] "+v"(ya0), [m0] "+v"(m0), [f0] "=&v"(f0), [y1] "+v"(ya1), [m1] "+v"(m1), [f1] "=&v"(f1), [e0] "=&s"(e0), [e1] "=&s"(e1), [ea] "=&s"(ea), [sv] "=&s"(sv)
                     : [c80] "v"(0x80u) : "vcc", "scc", "memory");
        y[0] = ya0 - jb; m[0] = m0; y[NS - 1u] = ya1 - jb; m[NS - 1u] = m1;
      }
#undef PE_HOP1
#undef PE_HOP2
    } else
#endif
    for (uint32_t h = 0; h < PE_HOPCAP; h++) {
      uint32_t f[NS]; bool any = false;
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) f[t] = lds_ld8(pb + PE_J1F + (m[t] != 0u ? y[t] : 0u));
      SC_STAGE();
      bool go[NS];
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { go[t] = m[t] != 0u && (f[t] & 0x80u) == 0u; any = any || go[t]; }
      if (__ballot(any) == 0ull) break;
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { y[t] += go[t] ? (f[t] & 15u) : 0u; m[t] -= go[t] ? 1u : 0u; }
    }
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) if (part[t]) n[t] = m[t];
  } else
  for (;;) {
    uint32_t f[NS], yc[NS], xb[NS]; bool go[NS]; bool any = false;
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      yc[t] = y[t] < c.Lp ? y[t] : 0u;
      xb[t] = 0u;
      if (J1) f[t] = lds_ld8(pb + PE_J1F + yc[t]);
      else { f[t] = ((lds_ld32(pb + PE_PM + ((yc[t] >> 5) << 2)) >> (yc[t] & 31u)) & 1u) << 7; xb[t] = pe_bits32(pb, yc[t]); }  // (the code word's bits in the same round trip as the flag)
    }
    SC_STAGE();
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      go[t] = ok[t] && n[t] != 0u && y[t] < c.Lp && (f[t] & 0x80u) == 0u && (!CAPPED || hops[t] < PE_HOPCAP);
      any = any || go[t];
    }
    if (__ballot(any) == 0ull) break;
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      uint32_t len;
      if (J1) len = f[t] & 15u;
      else { uint32_t sy; sc_lookup(c.lit_tree, xb[t], sy, len); }
      if (go[t]) { y[t] += len; n[t]--; hops[t]++; }
    }
  }
  uint32_t pmw[NS], cbw[NS], yc[NS]; bool inside[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    inside[t] = y[t] < c.Lp; yc[t] = inside[t] ? y[t] : 0u;
    pmw[t] = lds_ld32(pb + PE_PM + ((yc[t] >> 5) << 2)); cbw[t] = lds_ld16(pb + PE_CB + ((yc[t] >> 5) << 1));
  }
  SC_STAGE();
  uint32_t rk[NS], q2[NS]; bool onp[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    onp[t] = ((pmw[t] >> (yc[t] & 31u)) & 1u) != 0u;
    rk[t] = cbw[t] + (uint32_t)__builtin_popcount(pmw[t] & ((1u << (yc[t] & 31u)) - 1u));
    const uint32_t rr = rk[t] + n[t];
    q2[t] = lds_ld16(pb + PE_POR + ((rr < c.Rn ? rr : 0u) << 1));
  }
  SC_STAGE();
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    r[t].u = hops[t]; r[t].ry = rk[t]; r[t].hy = y[t]; r[t].hn = n[t];
    uint32_t code, next;
    if (!ok[t] || !inside[t]) { code = 2u; next = 0u; }
    else if (n[t] != 0u) {
      if (!onp[t]) { code = 3u; next = 0u; }
      else if (rk[t] + n[t] >= c.Rn) { code = 2u; next = 0u; }
      else if (imp[t]) { code = 1u; next = q2[t] | 0x8000u; }
      else { code = 0u; next = rk[t] + n[t]; }
    } else if (!imp[t] && onp[t]) { code = 0u; next = rk[t]; }
    else { code = 1u; next = y[t] | (imp[t] ? 0x8000u : 0u); }
    r[t].code = code; r[t].next = next;
  }
}

// The records' evaluation (two states a lane, side by side), written for the instruction count: the records are bound by the
// SIMDs' issue rate (tools/ubench/valu_rate.hip: one instruction per four clocks and SIMD whatever its kind), so everything a
// table can answer is a table -- PE_TD gives the bits of a whole distance code (ReadDistanceInternal, decode.rs:2066-2131:
// symbol + extra bits) where the distance tree gives the symbol, PE_TC what ReadCommandInternal (decode.rs:2134-2189) takes
// out of kCmdLut -- and a lane without a state evaluates bit 0 like everybody else instead of being masked out.
// In: d[t] = bit | kind << 15 of state t (0 where the lane has none), on[t]; a run that ran out of hops last time goes on from
// (ry[t], rn[t]) with rimp[t] (RES[t]).  Out: code[t] / next[t] as PeParse's; for code 3 ry / rn / rimp say where the run stands.
template <uint32_t NS>
__device__ __forceinline__ void pe_eval_rec(const PeCtx& c, const uint32_t (&d)[NS], const bool (&on)[NS], const bool (&res)[NS], uint32_t (&ry)[NS], uint32_t (&rn)[NS], uint32_t (&rimp)[NS],
                                            uint32_t (&code)[NS], uint32_t (&next)[NS]) {
  static_assert(NS == 1u || NS == 2u, "one or two states a lane");
  const uint32_t pb = c.pb;
  uint32_t q[NS], p[NS], kd[NS], lo[NS], hi[NS]; bool ok[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    const uint32_t pos = d[t] & 0x7FFFu;
    ok[t] = (bool)((uint32_t)on[t] & (uint32_t)(pos + 128u <= c.L)); q[t] = ok[t] ? pos : 0u; kd[t] = d[t] >> 15;
    lo[t] = pe_bits32(pb, q[t]);
  }
  SC_STAGE();
  {
    // the distance code at the state's bit: its length (a state of kind I has none: the read is harmless)
    uint32_t e[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) e[t] = lds_ld16(c.td + ((lo[t] & 0xFFu) << 1));
    SC_STAGE();
    if (__ballot((e[0] & 15u) > ROOT_BITS || (e[NS - 1u] & 15u) > ROOT_BITS) != 0ull) {
      uint32_t e2[NS];
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
        const uint32_t Ld = e[t] & 15u;
        const uint32_t idx = Ld > ROOT_BITS ? (e[t] >> 4) + __builtin_amdgcn_ubfe(lo[t], ROOT_BITS, Ld - ROOT_BITS) : (lo[t] & 0xFFu);
        e2[t] = lds_ld16(c.td + (idx << 1));
      }
      SC_STAGE();
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) e[t] = e2[t];   // (a leaf of the first level is read again: the same entry)
    }
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) p[t] = q[t] + (kd[t] == 0u ? e[t] >> 4 : 0u);
  }
  // the command's head
  uint32_t y[NS], n[NS], imp[NS];
  {
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) pe_bits64(pb, p[t], lo[t], hi[t]);
    SC_STAGE();
    uint32_t e[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) e[t] = lds_ld16(c.cmd_tree + ((lo[t] & 0xFFu) << 1));
    SC_STAGE();
    if (__ballot((e[0] & 15u) > ROOT_BITS || (e[NS - 1u] & 15u) > ROOT_BITS) != 0ull) {
      uint32_t e2[NS];
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
        const uint32_t Lh = e[t] & 15u;
        const uint32_t idx = Lh > ROOT_BITS ? (e[t] >> 4) + __builtin_amdgcn_ubfe(lo[t], ROOT_BITS, Lh - ROOT_BITS) : (lo[t] & 0xFFu);
        e2[t] = lds_ld16(c.cmd_tree + (idx << 1));
      }
      SC_STAGE();
      _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) if ((e[t] & 15u) > ROOT_BITS) e[t] = (e2[t] & ~15u) | ((e2[t] & 15u) + ROOT_BITS);   // (a code word is at most fifteen bits long)
    }
    uint32_t tc[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) tc[t] = lds_ld32(c.tc + ((e[t] >> 4) << 2));
    SC_STAGE();
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
      const uint32_t Lh = e[t] & 15u, ib = (tc[t] >> 15) & 31u, cb = (tc[t] >> 20) & 31u;
      const uint32_t xb = __builtin_amdgcn_alignbit(hi[t], lo[t], Lh);   // the bits behind the command symbol
      const uint32_t ins = (tc[t] & 0x7FFFu) + __builtin_amdgcn_ubfe(xb, 0u, ib);
      y[t] = p[t] + Lh + ib + cb; n[t] = ins; imp[t] = (tc[t] >> 25) & 1u;
      if (res[t]) { y[t] = ry[t]; n[t] = rn[t]; imp[t] = rimp[t]; }
    }
  }
  // the literal run, hop by hop through J1 until it is on the path (the bytes from Lp on carry the path flag: a run that
  // reaches them stops by itself), at most PE_HOPCAP hops
  {
    uint32_t m[NS]; bool part[NS];
    _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) { part[t] = (bool)((uint32_t)ok[t] & (uint32_t)(y[t] < c.Lp)); m[t] = part[t] ? n[t] : 0u; }
    const uint32_t jb = pb + PE_J1F;
    uint32_t ya0 = y[0] + jb, ya1 = y[NS - 1u] + jb, m0 = m[0], m1 = m[NS - 1u], f0, f1;
    uint64_t e0, e1, ea, sv;
    // (a state that has stopped hopping -- no literals left, or on the path -- stays stopped: the lanes still hopping are an
    // execution mask that only ever narrows; v_cmpx drops the lanes, the step itself is an add and a decrement)
#ifndef BROTLI_AMD_PE_REC_HOPS
#define BROTLI_AMD_PE_REC_HOPS 8
#endif
#if BROTLI_AMD_PE_REC_HOPS == 8
#define PE_HOPS_REC(H) H H H H H H H H
#elif BROTLI_AMD_PE_REC_HOPS == 6
#define PE_HOPS_REC(H) H H H H H H
#elif BROTLI_AMD_PE_REC_HOPS == 5
#define PE_HOPS_REC(H) H H H H H
#elif BROTLI_AMD_PE_REC_HOPS == 4
#define PE_HOPS_REC(H) H H H H
#elif BROTLI_AMD_PE_REC_HOPS == 12
#define PE_HOPS_REC(H) H H H H H H H H H H H H
#endif
#define PE_HOP1 \
      "s_mov_b64 exec, %[e0]\n\tds_read_u8 %[f0], %[y0]\n\ts_waitcnt lgkmcnt(0)\n\tv_cmpx_gt_u32 vcc, %[c80], %[f0]\n\tv_add_u32 %[y0], %[y0], %[f0]\n\t" \
      "v_subrev_u32 %[m0], 1, %[m0]\n\tv_cmpx_ne_u32 vcc, 0, %[m0]\n\ts_mov_b64 %[e0], exec\n\ts_cmp_eq_u64 %[e0], 0\n\ts_cbranch_scc1 .Lpe_hopr_done_%=\n\t"
#define PE_HOP2 \
      "s_mov_b64 exec, %[e0]\n\tds_read_u8 %[f0], %[y0]\n\ts_mov_b64 exec, %[e1]\n\tds_read_u8 %[f1], %[y1]\n\ts_mov_b64 exec, %[e0]\n\ts_waitcnt lgkmcnt(1)\n\t" \
      "v_cmpx_gt_u32 vcc, %[c80], %[f0]\n\tv_add_u32 %[y0], %[y0], %[f0]\n\tv_subrev_u32 %[m0], 1, %[m0]\n\tv_cmpx_ne_u32 vcc, 0, %[m0]\n\ts_mov_b64 %[e0], exec\n\t" \
      "s_mov_b64 exec, %[e1]\n\ts_waitcnt lgkmcnt(0)\n\t" \
      "v_cmpx_gt_u32 vcc, %[c80], %[f1]\n\tv_add_u32 %[y1], %[y1], %[f1]\n\tv_subrev_u32 %[m1], 1, %[m1]\n\tv_cmpx_ne_u32 vcc, 0, %[m1]\n\ts_mov_b64 %[e1], exec\n\t" \
      "s_or_b64 %[ea], %[e0], %[e1]\n\ts_cbranch_scc0 .Lpe_hopr_done_%=\n\t"
    if constexpr (NS == 1u) {
      asm volatile("s_mov_b64 %[sv], exec\n\tv_cmp_ne_u32 %[e0], 0, %[m0]\n\ts_cmp_eq_u64 %[e0], 0\n\ts_cbranch_scc1 .Lpe_hopr_done_%=\n\t"
                   PE_HOPS_REC(PE_HOP1)
                   ".Lpe_hopr_done_%=:\n\ts_mov_b64 exec, %[sv]"
                   : [y0] "+v"(ya0), [m0] "+v"(m0), [f0] "=&v"(f0), [e0] "=&s"(e0), [sv] "=&s"(sv)
                   : [c80] "v"(0x80u) : "vcc", "scc", "memory");
      (void)ya1; (void)m1; (void)f1; (void)e1; (void)ea;
      y[0] = ya0 - jb;
      if (part[0]) n[0] = m0;
    } else {
      asm volatile("s_mov_b64 %[sv], exec\n\tv_cmp_ne_u32 %[e0], 0, %[m0]\n\tv_cmp_ne_u32 %[e1], 0, %[m1]\n\ts_or_b64 %[ea], %[e0], %[e1]\n\ts_cbranch_scc0 .Lpe_hopr_done_%=\n\t"
                   PE_HOPS_REC(PE_HOP2)
                   ".Lpe_hopr_done_%=:\n\ts_mov_b64 exec, %[sv]"
                   : [y0] "+v"(ya0), [m0] "+v"(m0), [f0] "=&v"(f0), [y1] "+v"(ya1), [m1] "+v"(m1), [f1] "=&v"(f1), [e0] "=&s"(e0), [e1] "=&s"(e1), [ea] "=&s"(ea), [sv] "=&s"(sv)
                   : [c80] "v"(0x80u) : "vcc", "scc", "memory");
      y[0] = ya0 - jb; y[NS - 1u] = ya1 - jb;
      if (part[0]) n[0] = m0;
      if (part[NS - 1u]) n[NS - 1u] = m1;
    }
#undef PE_HOP1
#undef PE_HOP2
  }
  uint32_t pmw[NS], cbw[NS], yc[NS]; bool inside[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    inside[t] = y[t] < c.Lp; yc[t] = inside[t] ? y[t] : 0u;
    pmw[t] = lds_ld32(pb + PE_PM + ((yc[t] >> 5) << 2)); cbw[t] = lds_ld16(pb + PE_CB + ((yc[t] >> 5) << 1));
  }
  SC_STAGE();
  uint32_t rk[NS], q2[NS]; bool onp[NS];
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    onp[t] = ((pmw[t] >> (yc[t] & 31u)) & 1u) != 0u;
    rk[t] = cbw[t] + (uint32_t)__builtin_popcount(pmw[t] & ((1u << (yc[t] & 31u)) - 1u));
    const uint32_t rr = rk[t] + n[t];
    q2[t] = lds_ld16(pb + PE_POR + ((rr < c.Rn ? rr : 0u) << 1));
  }
  SC_STAGE();
  _Pragma("unroll") for (uint32_t t = 0; t < NS; t++) {
    // (flat on purpose: nested branches become execution-mask juggling, these are six selects)
    // (bitwise on purpose too: `&&` between two lanes' conditions comes out as a branch around the second one)
    const uint32_t live = (uint32_t)ok[t] & (uint32_t)inside[t], onp_ = (uint32_t)onp[t], more = (uint32_t)(n[t] != 0u) & (onp_ ^ 1u), over = onp_ & (uint32_t)(rk[t] + n[t] >= c.Rn), plain = onp_ & (imp[t] ^ 1u);
    // code: 2 where the state is not live; else 3 (more hops), 2 (beyond the ranks), 0 (a path state next), 1 (another state next)
    const uint32_t inner = more != 0u ? 3u : over != 0u ? 2u : 1u - plain;
    code[t] = live != 0u ? inner : 2u;
    next[t] = plain != 0u ? rk[t] + n[t] : (onp_ != 0u ? q2[t] : y[t]) | (imp[t] << 15);
    ry[t] = y[t]; rn[t] = n[t]; rimp[t] = imp[t];
  }
}
template <bool CAPPED, bool J1>
__device__ __forceinline__ PeParse pe_eval(const PeCtx& c, uint32_t pos, uint32_t kind, bool on) {
  const uint32_t pos_[1] = {pos}, kind_[1] = {kind}; const bool on_[1] = {on};
  PeParse r[1];
  pe_eval_n<1, CAPPED, J1>(c, pos_, kind_, on_, r);
  return r[0];
}


// A command with a long literal run gets regions of its own (wave 0, uniform): the region's path starts at the run's first
// literal, every path position is one of its literals, no records (ReadCommandInternal here, the literals in the run's
// regions, the distance and the copy in the checked loop: decode.rs:2134-2189, 2393-2462).  `hp` is the local bit of the
// command's head; on success the stream stands at the run's first literal.
#ifdef BROTLI_AMD_NO_TRYRUN
#define PE_TRY_RUN(st_, hp_) do { } while (0)
#define PE_TRY_RUN_FROM(st_, hp_, min_) do { } while (0)
#else
#define PE_TRY_RUN(st_, hp_) PE_TRY_RUN_FROM(st_, hp_, PE_RUN_MIN)
#define PE_TRY_RUN_FROM(st_, hp_, min_) do { \
    uint32_t lo_, hi_; \
    pe_bits64(pb, (hp_), lo_, hi_); \
    const ScHead h_ = sc_head(lo_, hi_, c.cmd_tree, c.lut_vgpr); \
    const uint32_t hins_ = rfl(h_.insert), hbits_ = rfl(h_.bits); \
    if (hins_ >= (min_) && hbits_ != 0u && (st_).bl1 != 0u && (rfl(h_.implicit) != 0u || (st_).bl2 != 0u)) { \
      (st_).run_on = 1u; (st_).run_rem = hins_; (st_).run_copy = rfl(h_.copy); (st_).run_implicit = rfl(h_.implicit); (st_).run_dctx = rfl(h_.dctx); \
      (st_).bl1 -= 1u; (st_).ncmd += 1u; (st_).b += hbits_; \
    } } while (0)
#endif

// The barrier of one engine's waves.  One engine a block: the hardware's.  Two: a counter in the engine's control words that
// only ever grows -- a wave adds one and waits until all GW have (s_barrier knows the block's waves only, and the other engine is
// somewhere else in its region).  A wave that waits unreasonably long stops the kernel rather than the machine.
#if PE_CFG_PIPE
__device__ __forceinline__ void pe_spin_check(uint32_t& spins) { if (++spins > (1u << 24)) __builtin_trap(); }
#define PE_SPIN_CHECK(s_) pe_spin_check(s_)
__device__ __forceinline__ void pe_gbar(const uint32_t pb, uint32_t& target) {
  target += GW;
  const uint32_t old = pe_atomic_add_uniform(pb + PE_CTL + 4u * PEC_GBAR, 1u);   // (waits for this wave's LDS traffic: lgkmcnt(0))
  if (old + 1u != target) { uint32_t spins = 0; while ((int32_t)(pe_ctl_ld(pb, PEC_GBAR) - target) < 0) { __builtin_amdgcn_s_sleep(1); pe_spin_check(spins); } }
}
#define PE_BAR() pe_gbar(pb, gb_target)
#else
#ifdef BROTLI_AMD_PROFILE_WAVES
// (profile: what every wave of block 0 spends between two barriers -- its ticks from the release of one to its arrival at the next,
// by the barrier's place in the source; the slowest wave of a step is the one the block waits for)
#ifndef BROTLI_AMD_WAVE_PROF_DEFINED
#define BROTLI_AMD_WAVE_PROF_DEFINED
}  // namespace
__device__ unsigned long long g_wave_prof[96][17];
namespace PE_CFG_NS {
#endif
#define PE_BAR() do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && lane == 0) { g_wave_prof[__COUNTER__ % 96][me] += t_ - wp_t; if (me == 0) g_wave_prof[(__COUNTER__ - 1) % 96][16] += 1; } \
                      __syncthreads(); wp_t = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PE_BAR() __syncthreads()
#endif
#if PE_CFG_REMOTE
// (a gang's waits inside an invocation are for another block's work on one region -- whose output may be hundreds of megabytes of overlapping
// copies: a minute or more of looking, not seconds, before the kernel is stopped rather than the machine)
__device__ __forceinline__ void pe_spin_check(uint32_t& spins) { if (++spins > (1u << 26)) __builtin_trap(); }
#define PE_SPIN_CHECK(s_) pe_spin_check(s_)
#else
#define PE_SPIN_CHECK(s_) do { } while (0)
#endif
#endif

// One invocation: every wave of the block calls it (wave 0 from process_commands, the others from helper_wave).
// Returns (wave 0) the number of commands it took; exit form and state in LDS_LEAN as the scan engine leaves them.
#if PE_DICT
// (PE_DICT) Wave 0, behind a pass that ended with the literals of a command whose copy is a word of the static dictionary: the
// word goes out behind them (decode.rs:2593-2640, as lean_rec_commands takes them) and the stream's state moves on; PEC_AGAIN
// says whether the commands behind it get a pass.  A function of its own: it is rare, and the engine's loops stay as they were.
__device__ __noinline__ void pe_dict_word(const uint32_t pbs, const uint32_t pb, gu8* const out, gcu8* const dict, const uint32_t m) {
  const uint32_t lane = lane_id();
  PeStream sw = pe_st_load(pbs);
  const uint32_t dd = pe_ctl_ld(pb, PEC_DICTD), wn_ = pe_ctl_ld(pb, PEC_DICTN), kd = pe_ctl_ld(pb, PEC_DICTK);
  const uint32_t maxd = sw.P < (uint64_t)(uint32_t)sw.max_backward ? (uint32_t)sw.P : (uint32_t)sw.max_backward;
  bool word = false; WordShape w = {}; uint32_t word_offset = 0;
  if (dd > maxd && wn_ >= 4u && wn_ <= 24u) {
    const uint32_t shift = kDictSizeBitsByLength[wn_];
    const uint32_t word_id = dd - maxd - 1u;
    const uint32_t transform_idx = word_id >> shift;
    if (transform_idx < (uint32_t)BROTLI_NUM_TRANSFORMS) {
      word_offset = kDictOffsetsByLength[wn_] + (word_id & mask_bits(shift)) * wn_;
      w = word_shape(wn_, transform_idx);
      word = w.total != 0u && w.total < sw.quota && (int32_t)w.total <= sw.mlen;
    }
  }
  uint32_t again = 0;
  if (word) {   // (the ring is not touched: decode.rs:2643-2644)
    const uint32_t ob = dictionary_word_bytes(dict, word_offset, w);
    if (lane < w.total) out[sw.P + lane] = (uint8_t)ob;
    sw.P += w.total; sw.quota -= w.total; sw.mlen -= (int32_t)w.total;
    pe_st_store(pbs, sw);
    again = (kd + 1u < m && sw.quota >= SC_MIN_QUOTA && sw.bl1 != 0u) ? 1u : 0u;
    if (again != 0u) { pe_ctl_st(pb, PEC_KS, kd + 1u); pe_ctl_st(pb, PEC_P0_LO, (uint32_t)sw.P); pe_ctl_st(pb, PEC_P0_HI, (uint32_t)(sw.P >> 32)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {   // not a plain word: the checked loop says what it is, from behind the command's distance on
    pe_ctl_st(pb, PEC_PDX, 1u); pe_ctl_st(pb, PEC_CONT, 0u);
  }
  pe_ctl_st(pb, PEC_AGAIN, again);
}

#endif
// A wave other than the decoding wave stays in here until the block is asked for something else than this engine: a call
// saves and restores the registers the caller may count on (38 vector registers a lane: 155 KB of scratch a block and
// invocation, more than a metablock's output is long; the lines are long out of L2 when the epilogue asks for them).  That, not
// the shape of the stores, was the WRITE_SIZE of 1.41 (long back-references) and 2.71 (high-entropy literals) times the
// output: 1.15 and 1.29 with the waves staying (tools/ubench/write_calib.hip: the counter is exact for every store pattern of
// this kernel).  The loop around the function's body costs the metric 1 % (the compiler's register allocation of the whole
// kernel shifts; -DBROTLI_AMD_PE_NO_STAY for the A/B).  Such a wave returns the number of the last request it has answered.
__device__ __noinline__ uint32_t path_engine(const uint32_t me_) {
pe_again:
  const uint32_t lane = lane_id();
  const uint32_t eng = PIPE2 ? rfl(me_) / GW : 0u;                // the engine this wave belongs to
  const uint32_t me = PIPE2 ? rfl(me_) % GW : rfl(me_);           // ... and its number in it
  const uint32_t T = PIPE2 ? threadIdx.x % (64u * GW) : threadIdx.x;
  const uint32_t pbs = hc_ld(HC_SCAN_BASE);                       // what the block's engines share
  const uint32_t pb = pbs + PE_SET0 + eng * PE_SET_BYTES;         // this engine's tables
  if (T == 0u) { lds_st32(pb + PE_CTL + 4u * PEC_NXOK, 0u); lds_st32(pb + PE_CTL + 4u * PEC_PDX, 0u); lds_st32(pb + PE_CTL + 4u * PEC_OVF, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DSEEN, 0u); }
  if (PIPE) {   // what the two engines tell each other starts from nothing
    if (threadIdx.x < 8u) lds_st32(pbs + PE_CTL + 4u * (PEC_RESOLVED + threadIdx.x), 0u);
    if (T == 0u) lds_st32(pb + PE_CTL + 4u * PEC_GBAR, 0u);
  }
  // ---- several CUs on one stream (PE_CFG_REMOTE; the control block's words: GC_* in brotli_kernels.hip) ----
  // The blocks of a gang take the stream's regions in turns as the two engines of a block do -- region k is block k mod gang's --, each with
  // the whole of its CU: the tables of a region (70 K clocks of the 130 K a region costs one CU) are built ahead by as many CUs as it takes,
  // and the stream itself only waits for the walk, the details and the resolve of the region before (and the execute for the output of the
  // region before).  The OWNER is the block that decodes the stream; it comes here from process_commands as ever and is member 0.  The HELPERS
  // live in this function: they wait for the owner's next invocation (EPOCH), take its parameters and the image of its table arena, do their
  // regions, say that they have left (READY) and wait again.
  uint32_t role = 0, gang_m = 1, epoch = 0; gu8* gc = nullptr; (void)role; (void)gang_m; (void)epoch; (void)gc;
  const uint64_t gs_t0 = __builtin_amdgcn_s_memtime(); (void)gs_t0;
  if (REMOTE) {
    // (HC_GANG_M: the gang's blocks; bit 8: a POOL launch -- nobody is dealt to a gang, a block whose own stream is done joins a stream that is
    // not, and a stream's gang is whoever has joined it when an invocation starts: see the kernel)
    role = hc_ld(HC_GANG_ROLE); gang_m = hc_ld(HC_GANG_M) & 0xFFu; gc = gang_ctl();
    const bool pool = (hc_ld(HC_GANG_M) >> 8) != 0u;
    const uint32_t seq_in = hc_ld(HC_SEQ);   // (a pool's helper goes back to its block's mailbox when the stream it helped is done)
    if (role == 0u) {
      epoch = hc_ld(HC_GANG_EPOCH) + 1u;   // (the word is this invocation's once it is everybody's: see below)
      if (pool && gang_m > 1u) {   // this invocation's gang: the helpers that have joined so far, seven at most; nobody: the one-block form's
        // (ONE look for the whole block: helpers join while it is taken)
        if (threadIdx.x == 0u) { const uint32_t joined = gang_ld32(gc, GC_JOINED); lds_st32(pbs + PE_CTL + 4u * PEC_MEMBERS, 1u + (joined < 7u ? joined : 7u)); }
        __syncthreads();
        gang_m = pe_ctl_ld(pbs, PEC_MEMBERS);
      }
      if (epoch >= (1u << 20) - 2u) {   // (the granules' tags hold twenty bits of it: a stream of a million invocations goes on without its gang)
        if (threadIdx.x == 0u) { gang_st32(gc, GC_EPOCH, GC_QUIT); *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * HC_GANG_M]) = 1u; }
        __syncthreads();
        gang_m = hc_ld(HC_GANG_M);
      } else if (epoch == 1u && !pool) {
        // the first time: have the helpers all started?  (They do so with the owner, give or take a microsecond; a block that is not running
        // cannot be waited for: the stream stays this block's alone then -- the mailbox says so from here on --, and a helper that turns up
        // finds the gang dissolved)
        if (threadIdx.x == 0u) {
          uint32_t tries = 0;
          while (gang_ld32(gc, GC_JOINED) < gang_m - 1u && tries < 2048u) { __builtin_amdgcn_s_sleep(16); tries++; }
          if (gang_ld32(gc, GC_JOINED) < gang_m - 1u) {
            gang_st32(gc, GC_EPOCH, GC_QUIT);
            *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * HC_GANG_M]) = 1u;
          }
        }
        __syncthreads();
        gang_m = hc_ld(HC_GANG_M);
      }
    }
    if (role == 0u && gang_m > 1u) {
      if (threadIdx.x == 0u) {   // the helpers have all left the invocation before (they read the image below when they enter one)
        uint32_t spins = 0; (void)spins;
        const uint64_t t0_ = __builtin_amdgcn_s_memtime(); (void)t0_;
        const uint32_t expected = *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * HC_GANG_READY]);   // (the helpers of every invocation so far, summed)
        while (gang_ld32(gc, GC_READY) != expected) { __builtin_amdgcn_s_sleep(8); PE_SPIN_CHECK(spins); }
        GANG_STAT(gc, 0, 1); GANG_STAT(gc, 5, __builtin_amdgcn_s_memtime() - t0_);
      }
      __syncthreads();
      const uint32_t ab = (pbs - LDS_FIXED + 15u) & ~15u;   // (the table arena lies between the fixed part and the engine's)
      for (uint32_t i = threadIdx.x << 4; i < ab; i += 64u * SC_WAVES * 16u)
        *reinterpret_cast<gu32x4*>(gc + GC_ARENA + i) = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[LDS_FIXED + i]);
      if (threadIdx.x < 32u) *reinterpret_cast<gu32*>(gc + GC_PARAMS + 4u * threadIdx.x) = lds_ld32(pbs + PE_CTL + 4u * threadIdx.x);
      else if (threadIdx.x < 40u) *reinterpret_cast<gu32*>(gc + GC_BR + 4u * (threadIdx.x - 32u)) = lds_ld32(LDS_BR + 4u * (threadIdx.x - 32u));
      else if (threadIdx.x == 40u) *reinterpret_cast<gu32*>(gc + GC_ARENA_BYTES) = ab;
      else if (threadIdx.x == 41u) gang_st64(gc, GC_MEMBERS, ((uint64_t)epoch << 32) | (uint64_t)gang_m);   // (with the invocation it is for: a pool's late comer must not take the next one's for this one's)
      gang_drain();
      __syncthreads();
      if (threadIdx.x == 0u) { gang_release(); GANG_STAT(gc, 18, __builtin_amdgcn_s_memtime() - gs_t0); }   // (the state, the plan and EPOCH follow below, where wave 0 has put the state together)
    } else if (role != 0u) {
      if (threadIdx.x == 0u) {
        uint32_t last = *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * HC_GANG_EPOCH]);
        uint32_t e;
        for (uint32_t idle = 0;; idle++) {   // (no cap: the owner may be busy with something else for as long as its stream takes)
          e = gang_ld32(gc, GC_EPOCH);
          if (e == GC_QUIT) break;
          if (e > last) {
            gang_acquire();
            // (a pool: an invocation that started before the owner had seen this block join is not this block's; nor is one that is over --
            // the word is the next one's already)
            const uint64_t mw = gang_ld64(gc, GC_MEMBERS);
            if ((uint32_t)(mw >> 32) == e && role < (uint32_t)mw) break;
            last = e;
          }
          if (idle < 4096u) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(127);
        }
        *reinterpret_cast<lds_vu32*>(&g_smem[LDS_HCTL + 4u * HC_GANG_EPOCH]) = e;
      }
      __syncthreads();
      epoch = hc_ld(HC_GANG_EPOCH);
      if (epoch == GC_QUIT) return seq_in;
      gang_m = (uint32_t)gang_ld64(gc, GC_MEMBERS);
      const uint32_t ab = *reinterpret_cast<gu32*>(gc + GC_ARENA_BYTES);
      for (uint32_t i = threadIdx.x << 4; i < ab && i < GC_ARENA_CAP; i += 64u * SC_WAVES * 16u)
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(&g_smem[LDS_FIXED + i]) = *reinterpret_cast<gu32x4*>(gc + GC_ARENA + i);
      if (threadIdx.x < 32u) lds_st32(pbs + PE_CTL + 4u * threadIdx.x, *reinterpret_cast<gu32*>(gc + GC_PARAMS + 4u * threadIdx.x));
      else if (threadIdx.x < 40u) lds_st32(LDS_BR + 4u * (threadIdx.x - 32u), *reinterpret_cast<gu32*>(gc + GC_BR + 4u * (threadIdx.x - 32u)));
    }
  }
  __syncthreads();  // the parameters are in place
#ifdef BROTLI_AMD_PROFILE_SCAN
  uint64_t pp_acc[32] = {}; uint64_t pp_t = __builtin_amdgcn_s_memtime();
#endif
  PeCtx c;
  c.pb = pb; c.td = pbs + PE_TD; c.tc = pbs + PE_TC;
  c.lit_tree = pe_ctl_ld(pbs, SCC_LIT_TREE); c.cmd_tree = pe_ctl_ld(pbs, SCC_CMD_TREE); c.dtree = pe_ctl_ld(pbs, SCC_DT0);
  c.postfix_bits = pe_ctl_ld(pbs, SCC_POSTFIX); c.num_direct = pe_ctl_ld(pbs, SCC_NUM_DIRECT);
  const uint32_t base_dw = pe_ctl_ld(pbs, SCC_BASE_DW), in_limit = pe_ctl_ld(pbs, SCC_IN_LIMIT);
  gu8* const out = (gu8*)(uintptr_t)((uint64_t)pe_ctl_ld(pbs, SCC_OUT_LO) | ((uint64_t)pe_ctl_ld(pbs, SCC_OUT_HI) << 32));
  gcu8* const dict = (gcu8*)(uintptr_t)((uint64_t)pe_ctl_ld(pbs, SCC_DICT_LO) | ((uint64_t)pe_ctl_ld(pbs, SCC_DICT_HI) << 32)); (void)dict;
  gcu32* const in_dw = BitReader::base() + base_dw;
  const uint32_t limit_dw = (in_limit + 31u) >> 5;
  c.lut_vgpr = 0;
  if (lane < 24) c.lut_vgpr = (uint32_t)kInsBase[lane] | ((uint32_t)kInsExtra[lane] << 16);
  else if (lane >= 32 && lane < 56) c.lut_vgpr = (uint32_t)kCopyBase[lane - 32] | ((uint32_t)kCopyExtra[lane - 32] << 16);

  // ---- wave 0: the stream's state (uniform), into its LDS words ----
  if (me == 0 && (!REMOTE || role == 0u)) {
    PeStream st;
    st.b = pe_ctl_ld(pbs, SCC_ENTRY);  // next command (bits from the engine's origin)
    st.P = (uint64_t)LEAN_LD(L_P_LO) | ((uint64_t)LEAN_LD(L_P_HI) << 32);
    st.quota = LEAN_LD(L_QUOTA); st.mlen = (int32_t)LEAN_LD(L_MLEN);
    st.bl0 = LEAN_LD(L_BL0); st.bl1 = LEAN_LD(L_BL1); st.bl2 = LEAN_LD(L_BL2);
    st.d0 = (int32_t)LEAN_LD(L_D0); st.d1 = (int32_t)LEAN_LD(L_D1); st.d2 = (int32_t)LEAN_LD(L_D2); st.d3 = (int32_t)LEAN_LD(L_D3);
    st.max_backward = (int32_t)LEAN_LD(L_MAX_BACKWARD);
    st.ncmd = 0;
    // a literal run that gets regions of its own: literals still to come, then the command's copy length, distance kind, context
    st.run_on = 0; st.run_rem = 0; st.run_copy = 0; st.run_implicit = 0; st.run_dctx = 0;
    st.rbl = PE_RBL;  // bits the next region takes: halved where the closure ran out of room, doubled back where it is small
    st.first = 1u;
    st.s_bits = 0u; st.s_cmds = 0u; st.s_lits = 0u; st.s_dsts = 0u;
    pe_st_store(pbs, st);
    bool long_first = false;
    if (REMOTE && st.b + 64u <= in_limit) {
      // (a gang) a first command whose literal run wants regions of its own is the one-block form's: seen here, in the stream's own bits,
      // nobody else hears of the invocation -- it costs the owner a look, not every block a region's tables
      const uint32_t d0_ = st.b >> 5, sh_ = st.b & 31u;
      const uint32_t a0_ = d0_ < limit_dw ? in_dw[d0_] : 0u, a1_ = d0_ + 1u < limit_dw ? in_dw[d0_ + 1u] : 0u, a2_ = d0_ + 2u < limit_dw ? in_dw[d0_ + 2u] : 0u;
      const ScHead h_ = sc_head(__builtin_amdgcn_alignbit(a1_, a0_, sh_), __builtin_amdgcn_alignbit(a2_, a1_, sh_), c.cmd_tree, c.lut_vgpr);
      long_first = rfl(h_.insert) >= PE_RUN_MIN && rfl(h_.bits) != 0u;
    }
    if (REMOTE && gang_m <= 1u) long_first = true;   // (the gang is dissolved, or a pool has sent nobody yet: the same way out)
    if (REMOTE) pe_ctl_st(pb, PEC_NOHELP, gang_m <= 1u && (hc_ld(HC_GANG_M) >> 8) != 0u ? 1u : 0u);
    if (REMOTE) pe_ctl_st(pb, PEC_PLAN, long_first ? 6u : 0u);
    if (REMOTE && long_first) pe_ctl_st(pbs, PEC_DECLINE, 1u);
    if (REMOTE && !long_first) {   // the invocation is everybody's: the stream's state in front of region 0, the plan (from region 0 on, at the entry), then its number
      hc_st(HC_GANG_EPOCH, epoch);
      hc_st(HC_GANG_READY, hc_ld(HC_GANG_READY) + gang_m - 1u);   // (what READY says when this invocation's helpers have all left it)
      lds_sync();
      const uint32_t v = lane < 25u ? *reinterpret_cast<lds_vu32*>(&g_smem[pbs + PE_CTL + 4u * (PEC_STATE + (lane < 25u ? lane : 0u))]) : lane == 25u ? 1u : 0u;
      if (lane < GC_STATE_WORDS) gang_st64(gc, GC_STATE + 8u * lane, (uint64_t)v | ((uint64_t)(epoch << 12) << 32));
      if (lane == 0u) { gang_st64(gc, GC_PLAN, (uint64_t)st.b); gang_st64(gc, GC_ENTRY, (uint64_t)st.b | ((uint64_t)(epoch << 12) << 32)); }
      gang_drain();
      if (lane == 0u) gang_st32(gc, GC_EPOCH, epoch);
    }
  }
  // ---- the records' tables (see pe_eval_rec) ----
  for (uint32_t i = threadIdx.x; i < 704u; i += 64u * SC_WAVES) {
    const uint32_t cell = i >> 6;
    const uint32_t ins_code = (((0x298500u >> (cell * 2)) & 3u) << 3) | ((i >> 3) & 7u);
    const uint32_t copy_code = (((0x262444u >> (cell * 2)) & 3u) << 3) | (i & 7u);
    lds_st32(c.tc + (i << 2), (uint32_t)kInsBase[ins_code] | ((uint32_t)kInsExtra[ins_code] << 15) | ((uint32_t)kCopyExtra[copy_code] << 20) | (i < 128u ? 1u << 25 : 0u));
  }
  if (threadIdx.x == 0u) lds_st32(pbs + PE_CTL + 4u * PEC_TDN, 256u);
  __syncthreads();
  if (threadIdx.x < 256u) {
    const uint32_t e = lds_ld16(c.dtree + (threadIdx.x << 1)), Ld = e & 15u;
    if (Ld > ROOT_BITS) __hip_atomic_fetch_max(reinterpret_cast<pe_lds_u32*>(&g_smem[pbs + PE_CTL + 4u * PEC_TDN]), (e >> 4) + (1u << (Ld - ROOT_BITS)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  const uint32_t td_n = pe_ctl_ld(pbs, PEC_TDN);
  for (uint32_t i = threadIdx.x; i < td_n && i < PE_TD_ENTRIES; i += 64u * SC_WAVES) {
    uint32_t e = lds_ld16(c.dtree + (i << 1));
    const uint32_t l4 = e & 15u;
    if (i >= 256u || l4 <= ROOT_BITS) {   // a leaf: the symbol's bits and its extra bits (decode.rs:2099-2128)
      const uint32_t code = e >> 4, Lw = l4 + (i >= 256u ? ROOT_BITS : 0u);
      const int32_t dv = (int32_t)code - (int32_t)c.num_direct;
      const uint32_t nb = (code >= 16u && dv >= 0) ? (((uint32_t)dv >> c.postfix_bits) >> 1) + 1u : 0u;
      e = l4 | ((Lw + nb) << 4);
    }
    lds_st16(c.td + (i << 1), e);
  }
  const bool td_ok = td_n <= PE_TD_ENTRIES;   // (a table that does not fit: the engine leaves the metablock to the one-wave loop)
  uint32_t pre_a = 0, pre_b = 0; bool pre_ok = false;  // the next region's input dwords of this lane, once they are known
  uint32_t lbdw = 0, le = 0, wn = 0; uint64_t P0 = 0;   // the region: its first dword, the entry's bit in it, its closure states, where its output starts
#ifdef BROTLI_AMD_PROFILE_WAVES
  uint64_t wp_t = __builtin_amdgcn_s_memtime(); (void)wp_t;
#endif
  uint32_t gb_target = 0; (void)gb_target;               // (two engines: this engine's barriers so far, times GW)
  uint32_t rseq = 0;                                     // regions of this invocation so far (the one at hand included)
  uint32_t kseq = 0; (void)kseq;                         // (two engines: the number of the region this engine is at)
  uint64_t gs_arr = 0; (void)gs_arr;                     // (gang statistics: when the stream arrived at this engine's region)
#ifdef BROTLI_AMD_GANG_TRACE   // (a gang: a line a region of invocation BROTLI_AMD_GANG_TRACE with wave 0's clock at every hand-over -- which chain binds?)
  uint64_t gt_ts[15] = {};
#define GT(k) do { gt_ts[k] = __builtin_amdgcn_s_memrealtime(); } while (0)   // (the 100 MHz clock all CUs share: s_memtime is a CU's own)
#define GTC(w, v) do { if (lane == 0) pe_atomic_add_uniform(pb + PE_CTL + 4u * (120u + (w)), (v)); } while (0)
#else
#define GT(k) do { } while (0)
#define GTC(w, v) do { } while (0)
#endif
#ifdef BROTLI_AMD_PROFILE_REGIONS
  uint64_t rg_ts[10] = {}; uint64_t rg_prev_end = 0; uint64_t rg_rt[4] = {}; uint32_t rg_rn[4] = {};
#define RG_STAMP(k) do { rg_ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_STAMP(k) do { } while (0)
#endif
  // What the region's tables start from (the engine's wave 0): the window and the counters of the phases.
  auto setup_tables = [&](const uint32_t lbdw_, const uint32_t le_, const uint32_t bits, const uint32_t mode, const uint32_t ent_) {
    pe_ctl_st(pb, PEC_LBDW, lbdw_); pe_ctl_st(pb, PEC_LE, le_); pe_ctl_st(pb, PEC_L, bits);
    pe_ctl_st(pb, PEC_WN, 1u); pe_ctl_st(pb, PEC_ON, 0u); pe_ctl_st(pb, PEC_NEXTRANK, 0u); pe_ctl_st(pb, PEC_TAILN, 0u); pe_ctl_st(pb, PEC_TAILNEXT, 0u); pe_ctl_st(pb, PEC_READY, 0u); pe_ctl_st(pb, PEC_TMIN, PE_CHUNKS);
    pe_ctl_st(pb, PEC_CHG, 0u); pe_ctl_st(pb, PEC_CHG + 1, 0u); pe_ctl_st(pb, PEC_CHG + 2, 0u);
    pe_ctl_st(pb, PEC_MODE, mode); pe_ctl_st(pb, PEC_ENT, ent_);
    if (REMOTE) pe_ctl_st(pb, PEC_SEEDN, 0u);
  };
  // (a gang, wave 0, behind setup_tables) entry seeds: n states from bit lo_ on (see PEC_SEEDLO)
  auto seed_entries = [&](const uint32_t lo_, const uint32_t n_) {
    pe_ctl_st(pb, PEC_SEEDLO, lo_); pe_ctl_st(pb, PEC_SEEDN, n_); pe_ctl_st(pb, PEC_WN, 1u + n_);
  };
  // ... and what the walk and what follows it start from: where the region's output begins, nothing published yet
  auto setup_walk = [&](const uint64_t P_) {
    pe_ctl_st(pb, PEC_NAPUB, 0u); pe_ctl_st(pb, PEC_WDONE, 0u); pe_ctl_st(pb, PEC_DCAND, 0u);
    pe_ctl_st(pb, PEC_P0_LO, (uint32_t)P_); pe_ctl_st(pb, PEC_P0_HI, (uint32_t)(P_ >> 32));
  };
  // n bytes out of LDS to memory in whole sixteen-byte lines: the bytes in front of the first line of `dst` and behind the last one
  // singly, every store between them aligned.  `src` is congruent to `dst` modulo sixteen -- whoever puts the bytes together in LDS
  // starts that many bytes in --, so a line of memory is a line of LDS.  (The stores of the round-3 write-out started wherever the
  // region's output did: WRITE_SIZE 1.4 to 2.5 times the bytes.)
  auto write_out = [&](const uint32_t src, gu8* const dst, const uint32_t n) {
    const uint32_t head0 = (16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u, head = head0 < n ? head0 : n, nb = (n - head) >> 4, tail0 = head + (nb << 4);
    if (T < head) dst[T] = (uint8_t)lds_ld8(src + T);
    if (T < n - tail0) dst[tail0 + T] = (uint8_t)lds_ld8(src + tail0 + T);
    for (uint32_t q = T; q < nb; q += 64u * GW)
      *reinterpret_cast<gu32x4*>(dst + head + ((uint64_t)q << 4)) = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[src + head + (q << 4)]);
  };
  // ================= a region of a long literal run (one engine) =================
  // Nothing but literals from an exactly known bit on: no tables per bit.  Lane t decodes the code words of bits 128 t .. 128 t + 127
  // one after the other (decode.rs:2393-2462) from where the word that straddles into its part ends -- a guess (0) at first, then
  // the lane before's exit, decoded again wherever that changed, until nothing does (prefix codes re-synchronise: two or three
  // rounds; a cap cuts the region in front of the first lane that has not settled).  Ranks by prefix sum; one more pass puts the
  // literals where they belong.  As many as the run, the literal block, the output limits and the region hold (one short of
  // every limit: what happens AT a limit is the checked loop's) go out; the next region starts behind them.
  auto run_region = [&]() -> uint32_t {
    const uint32_t ent = pe_ctl_ld(pb, PEC_ENT), et = ent / PE_RUN_SB;
    const uint32_t lim = c.L > 16u ? c.L - 16u : 0u;       // (a code word at or beyond may reach beyond the input)
    // the rest of the region's input (the first PE_CHUNKS + 6 dwords are there): asked for side by side, then stored
    { const uint32_t ndw = (c.L + 31u) / 32u + 6u;
      constexpr uint32_t NI = (PE_RUN_RBL / 32u + 6u - PE_CHUNKS - 6u + 64u * GW - 1u) / (64u * GW);
      uint32_t iv[NI];
      _Pragma("unroll") for (uint32_t q = 0; q < NI; q++) { const uint32_t i = PE_CHUNKS + 6u + T + q * 64u * GW; iv[q] = (i < ndw && lbdw + i < limit_dw) ? in_dw[lbdw + i] : 0u; }
      _Pragma("unroll") for (uint32_t q = 0; q < NI; q++) { const uint32_t i = PE_CHUNKS + 6u + T + q * 64u * GW; if (i < ndw) lds_st32(pb + PE_IN + (i << 2), iv[q]); } }
    PE_BAR();
    const uint32_t base = T * PE_RUN_SB;
    const bool act = T >= et && base < lim;
    uint32_t e = T == et ? ent % PE_RUN_SB : 0u, ex = 0, cnt = 0, np = 0;
    // the lane's code words from bit `e` of its part: how many, and where the last one ends; `emit`: the literals to their ranks
    // A table of the literal code by its first eleven bits -- symbol << 4 | length, 0 where eleven bits do not hold the code
    // word: one look-up a literal where the tree's two levels take two
    const uint32_t rl = pb + PE_RUN_LIT + ((uint32_t)P0 & 15u);   // (the literals in LDS as they lie in memory, modulo sixteen: see write_out)
    const uint32_t wt = pb + PE_RUN_EX + 64u * GW;   // (behind the lanes' exits, in the room of the path's chunk words)
    for (uint32_t i = T; i < 2048u; i += 64u * GW) { uint32_t sy, ln; sc_lookup(c.lit_tree, i, sy, ln); lds_st16(wt + (i << 1), ln <= 11u ? (sy << 4) | ln : 0u); }
    PE_BAR();
    // The lane's stream bits live in five registers (its 128 and the 32 behind them), moved down by every code word's length
    // The lane's stream bits live in five registers (its 128 and the 32 behind them), moved down by every code word's length
    // (`part`: the 256 bits the lane decodes -- its own, T, in the passes over the whole region; any, where only the parts whose entry
    // has moved are decoded again, by the block's first lanes)
    auto decode = [&](const bool on, const bool emit, const uint32_t rank0, const uint32_t want, const uint32_t part, const uint32_t e) {
      const uint32_t base = part * PE_RUN_SB;
      uint32_t y = e, k = 0;
      constexpr uint32_t NW = PE_RUN_SB / 32u + 1u;
      uint32_t w[NW];
      _Pragma("unroll") for (uint32_t j = 0; j < NW - 1u; j += 4u) {
        const u32x4 wv = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[pb + PE_IN + part * (PE_RUN_SB / 8u) + 4u * j]);
        w[j] = wv.x; w[j + 1u] = wv.y; w[j + 2u] = wv.z; w[j + 3u] = wv.w;
      }
      w[NW - 1u] = lds_ld32(pb + PE_IN + part * (PE_RUN_SB / 8u) + 4u * (NW - 1u));
      for (uint32_t q = e >> 5; __ballot(q != 0u) != 0ull; q = q != 0u ? q - 1u : 0u)   // (the run's first lane enters anywhere in its part)
        if (q != 0u) { _Pragma("unroll") for (uint32_t j = 0; j + 1u < NW; j++) w[j] = w[j + 1u]; w[NW - 1u] = 0u; }
      { const uint32_t r5 = e & 31u; _Pragma("unroll") for (uint32_t j = 0; j + 1u < NW; j++) w[j] = __builtin_amdgcn_alignbit(w[j + 1u], w[j], r5); w[NW - 1u] >>= r5; }
      for (;;) {
        const bool go = (bool)((uint32_t)on & (uint32_t)(y < PE_RUN_SB) & (uint32_t)(base + y < lim));
        if (__ballot(go) == 0ull) break;
        uint32_t ent = lds_ld16(wt + ((w[0] & 0x7FFu) << 1));
        if (__ballot((bool)((uint32_t)go & (uint32_t)(ent == 0u))) != 0ull) {   // (a code word of twelve bits and more: the tree's own two levels)
          uint32_t sy, ln; sc_lookup(c.lit_tree, w[0], sy, ln);
          ent = ent == 0u ? (sy << 4) | ln : ent;
        }
        const uint32_t ln = go ? ent & 15u : 0u;
        if (emit) { lds_st8(go ? rl + rank0 + k : pb + PE_CTL + 4u * PEC_SCRATCH, ent >> 4); if (go && rank0 + k == want) np = base + y; }
        _Pragma("unroll") for (uint32_t j = 0; j + 1u < NW; j++) w[j] = __builtin_amdgcn_alignbit(w[j + 1u], w[j], ln);
        w[NW - 1u] >>= ln;
        y += ln; k += go ? 1u : 0u;
      }
      if (on && !emit) { cnt = k; ex = y >= PE_RUN_SB ? y - PE_RUN_SB : 0u; }
    };
    PE_PROF(1);
    RG_STAMP(6);
    decode(act, false, 0u, 0u, T, e);
    RG_STAMP(7);
    PE_PROF(17);
    // The entries settle: a part whose entry is not where the part before it ends is decoded again from there.  The first such round
    // is nearly everybody's (the guesses were guesses) and goes lane by part as the first pass did; in the rounds behind it few parts
    // are left -- a high-entropy code re-synchronises slowly, one part in seven or so passes a wrong exit on -- and those go on a list
    // that the block's first lanes take, a part each: a round costs what its parts cost, not a pass of all sixteen waves (round 4:
    // three and a half passes' worth of rounds; the list leaves one and a bit).
    uint32_t tmin = 64u * GW;
    const uint32_t exa = pb + PE_RUN_EX, cna = pb + PE_RUN_CNT, ena = pb + PE_RUN_EX + 64u * GW + 4096u;   // u8 exit, u16 count, u8 entry per part
    lds_st8(exa + T, act ? ex : 0u); lds_st16(cna + (T << 1), act ? cnt : 0u); lds_st8(ena + T, e);
    for (uint32_t round = 0;; round++) {
      const uint32_t qn = pb + PE_CTL + 4u * (PEC_CHG + round % 3u);      // (a counter of three in turns: this round's list length)
      const uint32_t qa = pb + PE_RUN_QA + (round & 1u) * 64u * GW * 2u;
      if (T == 0u) lds_st32(pb + PE_CTL + 4u * (PEC_CHG + (round + 1u) % 3u), 0u);
      PE_BAR();
      // part T: does it start where part T - 1 ends?
      const uint32_t ne = T > et ? lds_ld8(exa + T - 1u) : lds_ld8(ena + T);
      const bool changed = (bool)((uint32_t)act & (uint32_t)(T > et) & (uint32_t)(ne != lds_ld8(ena + T)));
      // (bit 15 of a part's count: on this round's list -- a lane that goes on into the next part, below, stops in front of such a one)
      { const uint32_t cv = lds_ld16(cna + (T << 1)); lds_st16(cna + (T << 1), (cv & 0x7FFFu) | (changed ? 0x8000u : 0u)); }
      if (round >= 24u) { if (changed) pe_atomic_min(pb + PE_CTL + 4u * PEC_TMIN, T); PE_BAR(); tmin = pe_ctl_ld(pb, PEC_TMIN); break; }   // (a code that does not re-synchronise: the region ends where it has not)
      {
        const uint64_t cm = __ballot(changed);
        if (cm != 0ull) {
          uint32_t b0 = 0;
          if (lane == 0) b0 = pe_atomic_add(qn, (uint32_t)__popcll(cm));
          b0 = rfl(b0);
          if (changed) lds_st16(qa + ((b0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))) << 1), T | (ne << 11));
        }
      }
      PE_BAR();
      const uint32_t nq = rfl(lds_ld32(qn));
#ifdef BROTLI_AMD_PROFILE_REGIONS
      if (round < 4u) { rg_rt[round] = __builtin_amdgcn_s_memtime() - rg_ts[7]; rg_rn[round] = nq; }
#endif
      if (nq == 0u) break;
      if (nq >= 32u * GW) {   // most of the region: lane by part
        if (changed) lds_st8(ena + T, ne);
        decode(changed, false, 0u, 0u, T, ne);
        if (changed) { lds_st8(exa + T, ex); lds_st16(cna + (T << 1), cnt | 0x8000u); }
      } else if (T < nq) {   // the list, a part a lane (the waves behind its end have nothing to do)
        // ... and on into the parts behind it while the exit it finds is not the entry they were decoded from (nobody else's this
        // round: a part on the list is its lane's): what a moved exit sets off ends in this round instead of one round a part
        const uint32_t it = lds_ld16(qa + (T << 1));
        uint32_t pt = it & 2047u, pe_ = it >> 11; bool mine_ = true, listed = true;
        while (__ballot(mine_) != 0ull) {
          if (mine_) lds_st8(ena + pt, pe_);
          decode(mine_, false, 0u, 0u, mine_ ? pt : 0u, pe_);
          if (mine_) {
            lds_st8(exa + pt, ex); lds_st16(cna + (pt << 1), cnt | (listed ? 0x8000u : 0u));
            const uint32_t nx = pt + 1u;
            mine_ = nx < 64u * GW && nx * PE_RUN_SB < lim && lds_ld8(ena + nx) != ex && (lds_ld16(cna + (nx << 1)) & 0x8000u) == 0u;
            pt = nx; pe_ = ex; listed = false;
          }
        }
      }
    }
    // part T's entry and count, as they settled
    PE_BAR();
    e = lds_ld8(ena + T); cnt = act ? lds_ld16(cna + (T << 1)) & 0x7FFFu : 0u;
    PE_PROF(18);
    RG_STAMP(8);
    if (T >= tmin) cnt = 0;
    // ranks: exclusive prefix sum of the lanes' counts over the block; the region ends in front of the lane the literals' room runs out in
    uint32_t incl = sc_scan(cnt);
    if (lane == 63u) lds_st32(pb + PE_CTL + 4u * (PEC_WSUM + me), incl);
    PE_BAR();
    uint32_t wbase;
    { const uint32_t ws = lane < GW ? lds_ld32(pb + PE_CTL + 4u * (PEC_WSUM + lane)) : 0u; const uint32_t wi = sc_scan(ws); wbase = rdlane(wi - ws, me); }
    uint32_t cb = wbase + incl - cnt;
    if (cb + cnt > PE_RUN_LITCAP) pe_atomic_min(pb + PE_CTL + 4u * PEC_TMIN, T);
    PE_BAR();
    tmin = pe_ctl_ld(pb, PEC_TMIN);
    if (T >= tmin) cnt = 0;
    if (T == tmin || (tmin >= 64u * GW && T == 64u * GW - 1u)) *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_RN]) = T == tmin ? cb : cb + cnt;
    PE_BAR();
    const uint32_t Rn = pe_ctl_ld(pb, PEC_RN);
    PE_COUNT(22, Rn);
    if (me == 0) {
      const PeStream st = pe_st_load(pbs);
      uint32_t take = st.run_rem;
      const uint32_t cap1 = Rn != 0u ? Rn - 1u : 0u, cap2 = st.quota > 1u ? st.quota - 1u : 0u;
      take = take < cap1 ? take : cap1; take = take < st.bl0 ? take : st.bl0; take = take < cap2 ? take : cap2;
      pe_ctl_st(pb, PEC_TAKE, take);
#ifdef BROTLI_AMD_PE_DEBUG
      if (blockIdx.x == 0 && lane == 0) printf("run region: L %u entry %u Rn %u run_rem %u bl0 %u quota %u mlen %d -> take %u\n", c.L, ent, Rn, st.run_rem, st.bl0, st.quota, st.mlen, take);
#endif
    }
    PE_BAR();
    const uint32_t take = pe_ctl_ld(pb, PEC_TAKE);
    PE_PROF(2);
    RG_STAMP(9);
    // the literals to their ranks (and the bit of the first one that does not go out: where the stream goes on)
    decode(cnt != 0u && cb <= take, true, cb, take, T, e);
    if (cnt != 0u && cb <= take && take < cb + cnt) *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_NEXTRANK]) = np;
    PE_BAR();
    PE_PROF(4);
    write_out(rl, out + P0, take);
    if (me == 0) {
      PeStream st = pe_st_load(pbs);
      st.P += take; st.quota -= take; st.bl0 -= take; st.mlen -= (int32_t)take; st.run_rem -= take;
      const uint32_t npb = take != 0u ? pe_ctl_ld(pb, PEC_NEXTRANK) : ent;
      st.b = (lbdw << 5) + npb;
      pe_ctl_st(pb, PEC_CONT, (take != 0u && st.run_rem != 0u) ? 1u : 0u); pe_ctl_st(pb, PEC_NEXT_LBDW, st.b >> 5);
      pe_ctl_st(pb, PEC_FIN, (st.run_rem == 0u && npb + 96u <= c.L + 128u) ? 1u : 0u);
      PE_COUNT(19, take);
      pe_st_store(pbs, st);
    }
    PE_BAR();
    PE_PROF(5);
#ifndef BROTLI_AMD_PE_NO_RUN_FINISH
    if (pe_ctl_ld(pb, PEC_FIN) != 0u) {
      // The run is over: what is left of its command is a distance and a copy (decode.rs:2066-2131, 2583-2720).  Round 4 handed every
      // such command to the checked loop -- the invocation ended, the loop finished the command, and the engine came back for the next one
      // (a stream of high-entropy literals is nothing but such commands).  Now wave 0 takes the plain case in place -- a distance inside
      // the window, a copy inside every limit -- and the next region starts at the next command; anything else is the checked loop's as before.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the run's literals are in memory before the copy reads them)
      PE_BAR();
      if (me == 0) {
        PeStream st = pe_st_load(pbs);
        const uint32_t yb = st.b - (lbdw << 5);
        uint32_t dbits = 0, push = 0; int32_t dist = st.d0; bool ok_ = true;
        if (st.run_implicit == 0u) {
          uint32_t lo, hi;
          pe_bits64(pb, yb, lo, hi);
          const ScDist d = sc_dist(lo, hi, c.dtree, c.postfix_bits, c.num_direct);
          const uint32_t kd = rfl(d.kind), dv = rfl(d.val); dbits = rfl(d.bits);
          ok_ = st.bl2 != 0u && (st.b + dbits) <= in_limit;
          if (kd == SCK_EXPLICIT) { dist = (int32_t)dv; push = 1u; ok_ = ok_ && dv < (1u << 30); }
          else if (kd == SCK_SHORT) {
            if (dv != 0u) {   // TakeDistanceFromRingBuffer, decode.rs:2017-2049
              const uint32_t sh = dv << 1, back = 3u - ((0xaaafff1bu >> sh) & 3u);
              int32_t v = back == 0u ? st.d0 : back == 1u ? st.d1 : back == 2u ? st.d2 : st.d3;
              const int32_t mag = (int32_t)((0xfa5fa500u >> sh) & 3u);
              if (dv & 1u) v += mag; else v -= mag;
              dist = v; push = 1u;
            }
          } else ok_ = false;
        }
        const uint32_t n = st.run_copy;
        const uint32_t maxd = st.P < (uint64_t)(uint32_t)st.max_backward ? (uint32_t)st.P : (uint32_t)st.max_backward;
        ok_ = ok_ && dist > 0 && (uint32_t)dist <= maxd && n < st.quota && n < (1u << 24);
        if (ok_) {
          gu8* const dst = out + st.P; gu8* const src = dst - (uint32_t)dist;
          const uint32_t ud = (uint32_t)dist;
          if (ud < n) {   // the copy repeats itself (decode.rs:2657-2663, 2690-2720)
            if (ud >= 64u) { for (uint32_t q = lane; q < n + lane; q += 64u) if (q < n) dst[q] = src[q]; }
            else { uint32_t mm = lane % ud; const uint32_t step = 64u % ud; for (uint32_t q = 0; q < n; q += 64u) { if (q + lane < n) dst[q + lane] = src[mm]; mm += step; if (mm >= ud) mm -= ud; } }
          } else if (n <= 64u) { uint32_t t = 0; if (lane < n) t = src[lane]; if (lane < n) dst[lane] = (uint8_t)t; }
          else {
            const uint32_t n16 = n >> 4;
            for (uint32_t q = lane; q < n16; q += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)q * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)q * 16);
            const uint32_t tail = n16 << 4;
            if (tail + lane < n) dst[tail + lane] = src[tail + lane];
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st.P += n; st.quota -= n; st.mlen -= (int32_t)n;
          if (st.run_implicit == 0u) st.bl2 -= 1u;
          if (push != 0u) { st.d3 = st.d2; st.d2 = st.d1; st.d1 = st.d0; st.d0 = dist; }
          st.b += dbits;
          st.run_on = 0u; st.run_rem = 0u; st.run_copy = 0u; st.run_implicit = 0u; st.run_dctx = 0u;
          st.first = 1u;   // (the next region looks at its first command as an invocation's first region does: another long run, as a rule)
          st.s_cmds = 0u;
          pe_st_store(pbs, st);
          pe_ctl_st(pb, PEC_CONT, 1u); pe_ctl_st(pb, PEC_NEXT_LBDW, st.b >> 5);
        }
      }
      PE_BAR();
    }
#endif
    if (pe_ctl_ld(pb, PEC_CONT) == 0u) return 2u;
    {
      const uint32_t nl = pe_ctl_ld(pb, PEC_NEXT_LBDW);
      pre_a = nl + T < limit_dw ? in_dw[nl + T] : 0u; pre_b = (T < 6u && nl + PE_CHUNKS + T < limit_dw) ? in_dw[nl + PE_CHUNKS + T] : 0u;
      pre_ok = true;
    }
    return 1u;
  };
  // ================= the region's tables: input, J1, the path, the records, NEXT8 =================
  // (0: there they are; 1: the region was one of a long literal run and is done, on to the next; 2: the invocation ends)
  auto build = [&]() -> uint32_t {
    lbdw = pe_ctl_ld(pb, PEC_LBDW); le = pe_ctl_ld(pb, PEC_LE);
    c.L = pe_ctl_ld(pb, PEC_L);
    PE_COUNT(20, 1);
    // ---- input (asked for behind the resolve of the region before, where there was one) ----
    if (!pre_ok) { pre_a = lbdw + T < limit_dw ? in_dw[lbdw + T] : 0u; pre_b = (T < 6u && lbdw + PE_CHUNKS + T < limit_dw) ? in_dw[lbdw + PE_CHUNKS + T] : 0u; }
    lds_st32(pb + PE_IN + (T << 2), pre_a);
    if (T < 6u) lds_st32(pb + PE_IN + ((PE_CHUNKS + T) << 2), pre_b);
    pre_ok = false;
    PE_BAR();
    PE_PROF(0);
    if (!PIPE && me == 0 && pe_ctl_ld(pbs, PEC_STATE + 20) != 0u) {
      // the invocation's first region: is its first command one with a long literal run?  (later regions know from the resolve
      // of the region before)
      PeStream st = pe_st_load(pbs);
      st.first = 0u;
      if (c.L >= 256u) {
        PE_TRY_RUN(st, le);
        if (st.run_on != 0u) {
          // (the run's region takes four times an ordinary one's bits -- the window was laid out before anybody knew: round 4 ran every
          // run's first region with an ordinary region's 32 Kbit, 4 300 literals for what 34 700 cost)
          const uint32_t avail_ = in_limit - (lbdw << 5);
          pe_ctl_st(pb, PEC_MODE, 1u); pe_ctl_st(pb, PEC_ENT, st.b - (lbdw << 5)); pe_ctl_st(pb, PEC_L, avail_ < PE_RUN_RBL ? avail_ : PE_RUN_RBL);
        }
      }
      pe_st_store(pbs, st);
    }
#if !PE_CFG_PIPE && !PE_CFG_REMOTE && !defined(BROTLI_AMD_PE_OLD_RUN_REGIONS)
    PE_BAR();   // (the first region's mode is wave 0's word)
    if (pe_ctl_ld(pb, PEC_MODE) != 0u) { c.L = pe_ctl_ld(pb, PEC_L); return run_region(); }
#endif
    // ---- J1: the length of the literal code word at every bit, eight bits per lane and pass ----
#if defined(BROTLI_AMD_PE_REPEAT) && BROTLI_AMD_PE_REPEAT == 1
    for (int rep_ = 0; rep_ < 2; rep_++)
#endif
    for (uint32_t g = T; g < PE_RBL / 8u; g += 64u * GW) {
      const uint32_t pos0 = g << 3;
      const uint32_t v = pe_bits32(pb, pos0);
      uint32_t e[8], Lw[8];
      _Pragma("unroll") for (int j = 0; j < 8; j++) e[j] = lds_ld16(c.lit_tree + (((v >> j) & 0xFFu) << 1));
      _Pragma("unroll") for (int j = 0; j < 8; j++) Lw[j] = e[j] & 15u;
      bool any2 = false;
      _Pragma("unroll") for (int j = 0; j < 8; j++) any2 = any2 || Lw[j] > ROOT_BITS;
      if (__ballot(any2) != 0ull) {
        // (the eight second-level entries asked for side by side -- a lane that needs none reads the table's first entry --:
        // one round trip, not one per bit; codes of more than eight bits are the rule in data of high entropy)
        uint32_t e2[8];
        _Pragma("unroll") for (int j = 0; j < 8; j++) {
          const uint32_t idx = Lw[j] > ROOT_BITS ? (e[j] >> 4) + __builtin_amdgcn_ubfe(v >> j, ROOT_BITS, Lw[j] - ROOT_BITS) : 0u;
          e2[j] = lds_ld16(c.lit_tree + (idx << 1));
        }
        _Pragma("unroll") for (int j = 0; j < 8; j++) if (Lw[j] > ROOT_BITS) Lw[j] = ROOT_BITS + (e2[j] & 15u);
      }
      const uint32_t w0 = Lw[0] | (Lw[1] << 8) | (Lw[2] << 16) | (Lw[3] << 24), w1 = Lw[4] | (Lw[5] << 8) | (Lw[6] << 16) | (Lw[7] << 24);
      lds_st32(pb + PE_J1F + pos0, w0); lds_st32(pb + PE_J1F + pos0 + 4u, w1);
    }
    PE_BAR();
    PE_PROF(1);
    // ---- the path: chunk T's chain from its entry.  Inside a wave the entries settle through the lanes (a chunk's entry is
    // the exit of the chunk before: one cross-lane read a step, no barrier); between waves through LDS, a barrier a round.
    // A wave's entry is exact after as many rounds as waves lie in front of it, so PE_SYNC_ROUNDS rounds make the path exact
    // whatever the code (one that does not re-synchronise takes them all; the usual case is two).
    const uint32_t cbase = T << 5;
    const uint32_t ent = pe_ctl_ld(pb, PEC_ENT), et = ent >> 5, eoff = ent & 31u;  // the path's first bit: the entry, or a long run's first literal
    // The chunk's 32 code lengths in registers, and from them -- back to front, every index a constant -- where a chain
    // entering at each of its bits leaves it: xt nibble y = bits into the next chunk of the chain through bit y.
    uint32_t jw[8];
    {
      const u32x4 a = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[pb + PE_J1F + cbase]);
      const u32x4 bq = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(&g_smem[pb + PE_J1F + cbase + 16u]);
      jw[0] = a.x; jw[1] = a.y; jw[2] = a.z; jw[3] = a.w; jw[4] = bq.x; jw[5] = bq.y; jw[6] = bq.z; jw[7] = bq.w;
    }
    // (a window of sixteen nibbles slides down the chunk: nibble j = the exit of bit y + 1 + j, for the bits of the next
    // chunk what they are -- j bits into it; a code word is at most fifteen bits long, so the exit of bit y is one nibble of
    // the window, and the window behind bit y is the table's half)
    uint32_t xt[4];
    {
      uint64_t win = 0xFEDCBA9876543210ull;
      _Pragma("unroll") for (int y = 31; y >= 0; y--) {
        const uint32_t len = (jw[y >> 2] >> ((y & 3) * 8)) & 15u;
        const uint32_t exy = (uint32_t)(win >> (((len - 1u) & 15u) << 2)) & 15u;
        win = (win << 4) | exy;
        if (y == 16) { xt[2] = (uint32_t)win; xt[3] = (uint32_t)(win >> 32); }
      }
      xt[0] = (uint32_t)win; xt[1] = (uint32_t)(win >> 32);
    }
    auto exit_of = [&](uint32_t e) -> uint32_t {
      const uint32_t dsel = e < 8u ? xt[0] : e < 16u ? xt[1] : e < 24u ? xt[2] : xt[3];
      return (dsel >> ((e & 7u) * 4u)) & 15u;
    };
    uint32_t eo = T == et ? eoff : 0u, ex = exit_of(eo);
    uint32_t wave_entry = 0u, rounds = 0;
    for (;;) {
      for (;;) {  // the wave's own chunks: a chunk's entry is the exit of the chunk before
        const uint32_t prev_ex = bperm(((lane + 63u) & 63u) << 2, ex);
        const uint32_t neo = T == et ? eoff : lane == 0u ? wave_entry : prev_ex;
        const uint32_t nex = exit_of(neo);
        const bool changed = nex != ex;
        eo = neo; ex = nex;
        if (__ballot(changed) == 0ull) break;
      }
      // the wave's exit for the wave behind it; another round if any wave's entry moves
      const uint32_t slot = rounds & 1u;
      if (lane == 63u) lds_st8(pb + PE_EX + slot * 64u + me, ex);
      const uint32_t fw = pb + PE_CTL + 4u * (PEC_CHG + rounds % 3u);
      if (T == 0u) lds_st32(pb + PE_CTL + 4u * (PEC_CHG + (rounds + 1u) % 3u), 0u);
      PE_BAR();
      const uint32_t ne = me == 0u ? 0u : rfl(lds_ld8(pb + PE_EX + slot * 64u + me - 1u));
      if (ne != wave_entry && lane == 0) lds_st32(fw, 1u);
      wave_entry = ne;
      rounds++;
      PE_BAR();
      if (rfl(lds_ld32(fw)) == 0u) break;
      if (rounds >= PE_SYNC_ROUNDS) { if (T == 0u) lds_st32(pb + PE_CTL + 4u * PEC_TMIN, 0u); PE_BAR(); break; }  // (cannot happen: see above; no path, no region)
    }
    // the chunk's own path positions: the chain from its entry, out of the registers
    uint32_t pm = 0;
    {
      uint32_t y = eo;
      while (y < 32u) {
        pm |= 1u << y;
        const uint32_t k = y >> 2;
        const uint32_t d = k < 4u ? (k < 2u ? (k == 0u ? jw[0] : jw[1]) : (k == 2u ? jw[2] : jw[3])) : (k < 6u ? (k == 4u ? jw[4] : jw[5]) : (k == 6u ? jw[6] : jw[7]));
        y += (d >> ((y & 3u) * 8u)) & 15u;
      }
    }
    PE_COUNT(21, rounds);
    PE_PROF(17);
    // bits at or beyond L - 16 are not path positions (a code word there may reach beyond the input)
    {
      const uint32_t lim = c.L > 16u ? c.L - 16u : 0u;
      if (cbase + 32u > lim) pm = cbase >= lim ? 0u : pm & ((1u << (lim - cbase)) - 1u);
      if (T < et) pm = 0;  // (chunks in front of the path's first bit)
    }
    if (T >= pe_ctl_ld(pb, PEC_TMIN)) pm = 0;  // (only the entries' failure to settle writes it before this point -- in front of the loop's last barrier)
    // ranks: exclusive prefix sum of the chunks' counts over the block
    uint32_t cnt = (uint32_t)__builtin_popcount(pm);
    uint32_t incl = sc_scan(cnt);
    if (lane == 63u) lds_st32(pb + PE_CTL + 4u * (PEC_WSUM + me), incl);
    PE_BAR();
    uint32_t wbase = 0;
    {
      const uint32_t ws = lane < GW ? lds_ld32(pb + PE_CTL + 4u * (PEC_WSUM + lane)) : 0u;
      const uint32_t wi = sc_scan(ws);
      wbase = rdlane(wi - ws, me);
    }
    uint32_t cb = wbase + incl - cnt;
    if (cb + cnt > PE_RANKS) pe_atomic_min(pb + PE_CTL + 4u * PEC_TMIN, T);  // the ranks run out inside this chunk: the region ends in front of it
    PE_BAR();
    const uint32_t tmin = pe_ctl_ld(pb, PEC_TMIN);
    if (T >= tmin) { pm = 0; cnt = 0; }
    if (T == tmin || (tmin == PE_CHUNKS && T == PE_CHUNKS - 1u)) *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_RN]) = T == tmin ? cb : cb + cnt;
    lds_st32(pb + PE_PM + (T << 2), pm); lds_st16(pb + PE_CB + (T << 1), cb);
    PE_PROF(18);
    {
      // the chunk's path positions: bit, literal, flag -- two at a time, out of the chunk's 64 input bits
      const uint64_t w = (uint64_t)lds_ld32(pb + PE_IN + (T << 2)) | ((uint64_t)lds_ld32(pb + PE_IN + ((T + 1u) << 2)) << 32);
      uint32_t m = pm, r = cb;
      while (m != 0u) {
        const uint32_t o0 = (uint32_t)__builtin_ctz(m); m &= m - 1u;
        const bool two = m != 0u;
        const uint32_t o1 = two ? (uint32_t)__builtin_ctz(m) : o0; m &= m - 1u;
        const uint32_t x0 = (uint32_t)(w >> o0), x1 = (uint32_t)(w >> o1);
        uint32_t e0 = lds_ld16(c.lit_tree + ((x0 & 0xFFu) << 1)), e1 = lds_ld16(c.lit_tree + ((x1 & 0xFFu) << 1));
        uint32_t l0 = e0 & 15u, l1 = e1 & 15u;
        if (l0 > ROOT_BITS || l1 > ROOT_BITS) {
          const uint32_t i0 = l0 > ROOT_BITS ? (e0 >> 4) + __builtin_amdgcn_ubfe(x0, ROOT_BITS, l0 - ROOT_BITS) : (x0 & 0xFFu);
          const uint32_t i1 = l1 > ROOT_BITS ? (e1 >> 4) + __builtin_amdgcn_ubfe(x1, ROOT_BITS, l1 - ROOT_BITS) : (x1 & 0xFFu);
          const uint32_t f0 = lds_ld16(c.lit_tree + (i0 << 1)), f1 = lds_ld16(c.lit_tree + (i1 << 1));
          if (l0 > ROOT_BITS) { e0 = f0; l0 = ROOT_BITS + (f0 & 15u); }
          if (l1 > ROOT_BITS) { e1 = f1; l1 = ROOT_BITS + (f1 & 15u); }
        }
        lds_st16(pb + PE_POR + (r << 1), cbase + o0); lds_st8(pb + PE_J1F + cbase + o0, l0 | 0x80u); lds_st8(pb + PE_LIT + r, e0 >> 4);
        if (two) { lds_st16(pb + PE_POR + ((r + 1u) << 1), cbase + o1); lds_st8(pb + PE_J1F + cbase + o1, l1 | 0x80u); lds_st8(pb + PE_LIT + r + 1u, e1 >> 4); }
        r += two ? 2u : 1u;
      }
    }
    if (T == 0u) lds_st16(pb + PE_WST, le | 0x8000u);  // the closure's first state: a command starts at the entry (lane 0 evaluates it)
    const uint32_t seed_n = REMOTE ? pe_ctl_ld(pb, PEC_SEEDN) : 0u;
    if (REMOTE && T < seed_n) lds_st16(pb + PE_WST + ((1u + T) << 1), (pe_ctl_ld(pb, PEC_SEEDLO) + T) | 0x8000u);   // ... and the entry seeds behind it
    if (T == 1u) lds_st16(pb + PE_NEXT + (PE_STATES << 1), PEN_NONE);  // (NEXT8's sentinel)
    {
      const uint32_t lim = c.L > 16u ? c.L - 16u : 0u, cut = tmin << 5;
      c.Lp = lim < cut ? lim : cut;
    }
    // sentinels: the sixteen bytes of J1 from Lp on carry the path flag, so that the records' hop loops stop there by themselves
    // (no path position lies there: nobody else writes them)
    if (T < 16u) lds_st8(pb + PE_J1F + c.Lp + T, lds_ld8(pb + PE_J1F + c.Lp + T) | 0x80u);
    PE_BAR();
    c.Rn = pe_ctl_ld(pb, PEC_RN);
    PE_PROF(2);
    PE_COUNT(22, c.Rn);
    if (pe_ctl_ld(pb, PEC_MODE) != 0u) {
      // ---- a region of a long literal run: the path's literals, as many as the run, the literal block, the output limits and
      // the region hold (one short of each limit: what happens AT a limit is the checked loop's), go out; the next region
      // starts behind them ----
      if (me == 0) {
        const PeStream st = pe_st_load(pbs);
        uint32_t take = st.run_rem;
        const uint32_t cap1 = c.Rn != 0u ? c.Rn - 1u : 0u, cap2 = st.quota > 1u ? st.quota - 1u : 0u;
        take = take < cap1 ? take : cap1; take = take < st.bl0 ? take : st.bl0; take = take < cap2 ? take : cap2;
        pe_ctl_st(pb, PEC_TAKE, take);
      }
      PE_BAR();
      const uint32_t take = pe_ctl_ld(pb, PEC_TAKE);
      {
        gu8* const o = out + P0;
        for (uint32_t i = T << 2; i < take; i += 4u * 64u * GW) {
          const uint32_t v = lds_ld32(pb + PE_LIT + i);
          if (i + 4u <= take) *reinterpret_cast<gu32*>(o + i) = v;
          else { o[i] = (uint8_t)v; if (i + 1u < take) o[i + 1u] = (uint8_t)(v >> 8); if (i + 2u < take) o[i + 2u] = (uint8_t)(v >> 16); }
        }
      }
      if (me == 0) {
        PeStream st = pe_st_load(pbs);
        st.P += take; st.quota -= take; st.bl0 -= take; st.mlen -= (int32_t)take; st.run_rem -= take;
        const uint32_t np = take != 0u ? rfl(lds_ld16(pb + PE_POR + (take << 1))) : ent;
        st.b = (lbdw << 5) + np;
        pe_ctl_st(pb, PEC_CONT, (take != 0u && st.run_rem != 0u) ? 1u : 0u); pe_ctl_st(pb, PEC_NEXT_LBDW, st.b >> 5);
        PE_COUNT(19, take);
        pe_st_store(pbs, st);
      }
      PE_BAR();
      if (pe_ctl_ld(pb, PEC_CONT) == 0u) return 2u;
      {
        const uint32_t nl = pe_ctl_ld(pb, PEC_NEXT_LBDW);
        pre_a = nl + T < limit_dw ? in_dw[nl + T] : 0u; pre_b = (T < 6u && nl + PE_CHUNKS + T < limit_dw) ? in_dw[nl + PE_CHUNKS + T] : 0u;
        pre_ok = true;
      }
      return 1u;
    }
    // ---- records: every lane keeps two evaluations going side by side.  A lane that is through with a state takes the next
    // path position (kind E) off a shared counter; a lane whose record leads to a state that is not a path state (the run
    // ended before it met the path, or the command has an implicit distance) appends that state to the closure and
    // evaluates it itself next; a lane whose run used up its hops goes on with it next time.  No rounds, no barriers: the
    // loop ends when the counter is exhausted and no lane has anything left.
    // Two phases of the same loop (two states a lane, then one): the bulk on all waves -- until the counter is exhausted and a wave has less than a quarter
    // of its slots busy; what it still holds then (chains of states that are not path states, a few per wave) goes on a
    // list --, and the thin end of it with ONE state a lane (half the instructions a pass: what is left are chains, a pass
    // is one link of each), the waves taking the list's states the way the bulk took path positions.  (Measured: handing
    // over below 80 busy slots of 128, all sixteen waves in the second phase: 7.64 ms against 7.92 with 32 / four waves.)
    // ---- records: a persistent loop.  Source 0: every path position as kind E (a shared counter hands them out); a lane
    // whose record leads to a state that is not a path state (the run ended before it met the path, or the command has an implicit
    // distance) appends that state to the closure and evaluates it itself next; a lane whose run used up its hops goes on with it
    // next time.  Two states a lane until the counter is exhausted and a wave has less than PE_TAIL_AT of its slots busy -- what it
    // still holds (chains of states that are not path states) goes on the list.  Source 1: the list, one state a lane (half the
    // instructions a pass: what is left are chains, a pass is one link of each).  (Tried in round 4: a first pass over the path
    // positions without the bookkeeping, the closure in a second one -- slower, 8.1 against 7.2 ms: the closure's chains are deep
    // and thin, and only side by side with fresh path positions do they find the lanes busy.)
    auto records_loop = [&](auto nsl_, const uint32_t source, const uint32_t limit) {
      constexpr uint32_t NSL = decltype(nsl_)::value;
      uint32_t sid[NSL], dsc[NSL], ry[NSL], rn[NSL], rimp[NSL]; bool has[NSL], res[NSL];
      _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) { sid[t] = 0u; dsc[t] = 0u; ry[t] = 0u; rn[t] = 0u; rimp[t] = 0u; has[t] = false; res[t] = false; }
      if (source == 0u && T == 0u) { has[0] = true; sid[0] = PE_RANKS; dsc[0] = le | 0x8000u; }  // the closure's first state: a command starts at the entry
      uint32_t iters = 0; (void)iters;
      bool dry = false;  // the source has nothing more for this wave
      for (;;) {
        {  // free slots take new states off the source: one LDS atomic per wave and pass for all of them
          uint64_t nm[NSL]; uint32_t want = 0;
          _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) { nm[t] = __ballot(!has[t]); want += (uint32_t)__popcll(nm[t]); }
          if (want != 0u && !dry) {
            uint32_t base = pe_atomic_add_uniform(pb + PE_CTL + 4u * (source == 0u ? (uint32_t)PEC_NEXTRANK : (uint32_t)PEC_TAILNEXT), want);
            if (base + want > limit) dry = true;
            _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) {
              const uint32_t rr = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm[t] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm[t], 0u));
              base += (uint32_t)__popcll(nm[t]);
              bool take = (bool)((uint32_t)!has[t] & (uint32_t)(rr < limit));
              uint32_t idv = rr;
              if (REMOTE && source == 0u && rr >= c.Rn) idv = PE_RANKS + 1u + (rr - c.Rn);   // (behind the path positions: the entry seeds, closure states 1 ..)
              if (source == 1u) { idv = lds_ld16(pb + PE_TAILQ + ((take ? rr : 0u) << 1)); take = (bool)((uint32_t)take & (uint32_t)(idv < PEN_FIRST_SPECIAL)); }
              const uint32_t idc = take ? idv : 0u;
              const uint32_t stv = lds_ld16(pb + (idc < PE_RANKS ? PE_POR + (idc << 1) : PE_WST + ((idc - PE_RANKS) << 1)));
              has[t] = (bool)((uint32_t)has[t] | (uint32_t)take); res[t] = (bool)((uint32_t)res[t] & (uint32_t)!take);
              sid[t] = take ? idc : sid[t]; dsc[t] = take ? stv : dsc[t];
            }
          }
        }
        const uint64_t h0 = __ballot(has[0]), h1 = NSL > 1u ? __ballot(has[NSL - 1u]) : 0ull;
        if ((h0 | h1) == 0ull) break;
        if (source == 0u && dry && (uint32_t)__popcll(h0) + (uint32_t)__popcll(h1) < PE_TAIL_AT) {
          // the thin end: what this wave still holds goes on the list
          const uint32_t cnt = (uint32_t)__popcll(h0) + (uint32_t)__popcll(h1);
          uint32_t base = 0;
          if (lane == 0) base = pe_atomic_add(pb + PE_CTL + 4u * PEC_TAILN, cnt);
          base = rfl(base);
          if (base + cnt <= PE_TAILCAP) {
            const uint32_t i0 = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(h0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)h0, 0u));
            const uint32_t i1 = base + (uint32_t)__popcll(h0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(h1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)h1, 0u));
            // (a run that stands in the middle of its hops starts over on the list's side: the list holds states)
            if (has[0]) lds_st16(pb + PE_TAILQ + (i0 << 1), sid[0]);
            if (NSL > 1u && has[NSL - 1u]) lds_st16(pb + PE_TAILQ + (i1 << 1), sid[NSL - 1u]);
            break;
          }
          // (the list is full: this wave sees its states through itself; its claim on the list holds no states -- marked so)
          for (uint32_t i = base + lane; i < base + cnt && i < PE_TAILCAP; i += 64u) lds_st16(pb + PE_TAILQ + (i << 1), PEN_NONE);
        }
        uint32_t code[NSL], nxt[NSL];
        {
          uint32_t dd[NSL];
          _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) dd[t] = has[t] ? dsc[t] : 0u;
          pe_eval_rec<NSL>(c, dd, has, res, ry, rn, rimp, code, nxt);
        }
        iters++;
        {
          bool app[NSL]; uint64_t am[NSL]; uint32_t slot[NSL]; uint32_t wantw = 0;
          _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) { app[t] = (bool)((uint32_t)has[t] & (uint32_t)(code[t] == 1u)); am[t] = __ballot(app[t]); wantw += (uint32_t)__popcll(am[t]); slot[t] = 0; }
          if (wantw != 0u) {  // (appending closure states: one LDS atomic per wave and pass)
            uint32_t base = pe_atomic_add_uniform(pb + PE_CTL + 4u * PEC_WN, wantw);
            _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) {
              slot[t] = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(am[t] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am[t], 0u));
              base += (uint32_t)__popcll(am[t]);
            }
          }
          _Pragma("unroll") for (uint32_t t = 0; t < NSL; t++) {
            const bool fin = (bool)((uint32_t)has[t] & (uint32_t)(code[t] != 3u));                  // the record is there (code 3: more hops next time, from ry / rn)
            const bool goes_on = (bool)((uint32_t)app[t] & (uint32_t)(slot[t] < PE_WCAP));           // ... and leads to a state that is not a path state: this lane's next
            PE_LANECOUNT(30, app[t] && slot[t] >= PE_WCAP);
            const uint32_t nx = goes_on ? PE_RANKS + slot[t] : code[t] == 0u ? nxt[t] : code[t] == 2u ? (uint32_t)PEN_END : (uint32_t)PEN_BYHAND;
            // (no masks: a lane with nothing to store writes the scratch word)
            lds_st16(goes_on ? pb + PE_WST + (slot[t] << 1) : pb + PE_CTL + 4u * PEC_SCRATCH, nxt[t]);
            lds_st16(fin ? pb + PE_NEXT + (sid[t] << 1) : pb + PE_CTL + 4u * PEC_SCRATCH, nx);
            res[t] = (bool)((uint32_t)has[t] & (uint32_t)(code[t] == 3u));
            has[t] = (bool)((uint32_t)res[t] | (uint32_t)goes_on);
            sid[t] = goes_on ? PE_RANKS + slot[t] : sid[t];
            dsc[t] = goes_on ? nxt[t] : dsc[t];
          }
        }
      }
      PE_COUNT(23 - 12 * source, iters);
    };
#ifndef BROTLI_AMD_PE_BULK_NS
#define BROTLI_AMD_PE_BULK_NS 2
#endif
    records_loop(std::integral_constant<uint32_t, BROTLI_AMD_PE_BULK_NS>{}, 0u, c.Rn + seed_n);
    PE_BAR();
    {
      const uint32_t tail_n = pe_ctl_ld(pb, PEC_TAILN) < PE_TAILCAP ? pe_ctl_ld(pb, PEC_TAILN) : PE_TAILCAP;
      if (me < PE_TAIL_WAVES && tail_n != 0u) records_loop(std::integral_constant<uint32_t, 1>{}, 1u, tail_n);
    }
    PE_BAR();
    PE_COUNT(24, pe_ctl_ld(pb, PEC_WN) < PE_WCAP ? pe_ctl_ld(pb, PEC_WN) : PE_WCAP);
    wn = pe_ctl_ld(pb, PEC_WN) < PE_WCAP ? pe_ctl_ld(pb, PEC_WN) : PE_WCAP;
    if (me == 0) {
      const uint32_t raw = pe_ctl_ld(pb, PEC_WN); uint32_t rbl = pe_ctl_ld(pbs, PEC_STATE + 8);
      if (raw > PE_WCAP - PE_WCAP / 8u) {
        rbl = rbl > 8192u ? rbl >> 1 : rbl;
        // (a stream of few literals -- an executable: short copies one after the other -- has few path positions and long chains of
        // states beside the path; three such regions and the invocation ends with word to the caller: the scan engine's kind of
        // stream, a parse at every bit and no chains.  256 x libc.so.6 at -q 5: 0.39 against 0.77 G commands/s)
        pe_ctl_st(pb, PEC_OVF, pe_ctl_ld(pb, PEC_OVF) + 1u);
      } else if (raw < PE_GROW_BELOW && rbl < PE_RBL) rbl <<= 1;
      pe_ctl_st(pbs, PEC_STATE + 8, rbl);
    }
    PE_PROF(4);
    // ---- NEXT8: the state eight commands on (PEN_NONE where the way there is not all records) ----
#if defined(BROTLI_AMD_PE_REPEAT) && BROTLI_AMD_PE_REPEAT == 4
    for (int rep_ = 0; rep_ < 2; rep_++)
#endif
    // (twelve states a lane side by side -- a region's states in one go, as a rule --: the phase is eight dependent LDS round
    // trips whatever the number of states a lane carries through them)
    for (uint32_t j0 = T; j0 < c.Rn + wn; j0 += 12u * 64u * GW) {
      constexpr uint32_t NW = 12u;
      uint32_t a[NW], ix[NW];
      _Pragma("unroll") for (uint32_t t = 0; t < NW; t++) {
        const uint32_t j = j0 + t * 64u * GW;
        ix[t] = j < c.Rn ? j : j < c.Rn + wn ? PE_RANKS + (j - c.Rn) : PE_STATES;   // the path states, then the closure's
        a[t] = ix[t];
      }
      _Pragma("unroll") for (int h = 0; h < 8; h++) {
        // (a record that is no way on -- END, BYHAND -- is a number beyond the states: clamped, it reads the word behind the table,
        // which says NONE, and stays there)
        uint32_t v[NW];
        _Pragma("unroll") for (uint32_t t = 0; t < NW; t++) v[t] = lds_ld16(pb + PE_NEXT + ((a[t] < PE_STATES ? a[t] : PE_STATES) << 1));
        _Pragma("unroll") for (uint32_t t = 0; t < NW; t++) a[t] = v[t];
      }
      // (written behind a barrier: J1's room is read by nobody any more, the records are complete)
      _Pragma("unroll") for (uint32_t t = 0; t < NW; t++)
        if (ix[t] < PE_STATES) lds_st16(pb + PE_N8 + (ix[t] << 1), a[t] < PEN_FIRST_SPECIAL ? a[t] : (uint32_t)PEN_NONE);
    }
    PE_BAR();
    if (PE_JUMP_LOG == 4) {
      // ... and from it the state sixteen commands on, in place: every thread its own states' (one in 1024), read before the
      // barrier, written behind it
      constexpr uint32_t PER = (PE_STATES + 64u * GW - 1u) / (64u * GW);
      uint32_t b2[PER];
      _Pragma("unroll") for (uint32_t t = 0; t < PER; t++) {
        const uint32_t i = T + t * 64u * GW;
        const bool valid = i < c.Rn || (i >= PE_RANKS && i < PE_RANKS + wn);
        b2[t] = valid ? lds_ld16(pb + PE_N8 + (i << 1)) : (uint32_t)PEN_NONE;
      }
      _Pragma("unroll") for (uint32_t t = 0; t < PER; t++) {
        const uint32_t w2 = lds_ld16(pb + PE_N8 + ((b2[t] < PEN_FIRST_SPECIAL ? b2[t] : 0u) << 1));
        b2[t] = b2[t] < PEN_FIRST_SPECIAL ? w2 : (uint32_t)PEN_NONE;
      }
      PE_BAR();
      _Pragma("unroll") for (uint32_t t = 0; t < PER; t++) {
        const uint32_t i = T + t * 64u * GW;
        if (i < c.Rn || (i >= PE_RANKS && i < PE_RANKS + wn)) lds_st16(pb + PE_N8 + (i << 1), b2[t] < PEN_FIRST_SPECIAL ? b2[t] : (uint32_t)PEN_NONE);
      }
      PE_BAR();
    }
    PE_PROF(5);
    return 0u;
  };
  // (a gang, wave 0) The stream's state as the region before's resolve left it -- or the word that it will not come: the invocation ended in front
  // of this region (a limit cut the region before short of what its walk had listed: what was walked here was walked for nothing).  `walked`:
  // the region has been walked from the entry its state should confirm; otherwise the region cannot be taken at all (no input left, no table
  // for the distances) and the state is only waited for to say so: STOP counts the regions RESOLVED, and is written when that number is final --
  // by the resolve that ends the invocation, or here, behind the state of the last region that was resolved.
  auto full_arrival = [&](const bool walked) {
    const uint32_t want = (epoch << 12) | kseq;
    uint64_t v; uint32_t spins = 0; bool arrived, stopped; (void)spins;
    const uint64_t t0_ = __builtin_amdgcn_s_memtime(); (void)t0_;
    for (;;) {
      v = gang_ld64(gc, lane < GC_STATE_WORDS ? GC_STATE + 8u * lane : GC_STOP);
      arrived = __ballot(lane < GC_STATE_WORDS && (uint32_t)(v >> 32) == want) == ((1ull << GC_STATE_WORDS) - 1ull);
      stopped = rdlane((uint32_t)(v >> 32), 32) == epoch && rdlane((uint32_t)v, 32) <= kseq;
      if (arrived || stopped) break;
      __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins);
    }
    GANG_STAT(gc, 39, __builtin_amdgcn_s_memtime() - t0_);
    uint32_t ok_ = 0u;
    if (arrived && (rdlane((uint32_t)v, 25) & 1u) != 0u) {   // (the region before's resolve said that the stream goes on)
      if (lane < 25u) lds_st32(pbs + PE_CTL + 4u * (PEC_STATE + lane), (uint32_t)v);   // the state into this engine's own words
      lds_sync();
      const PeStream st = pe_st_load(pbs);
      ok_ = (walked && st.b == pe_ctl_ld(pb, PEC_MYENTRY) && st.quota >= SC_MIN_QUOTA && st.bl1 != 0u) ? 1u : 0u;
      pe_ctl_st(pb, PEC_P0_LO, (uint32_t)st.P); pe_ctl_st(pb, PEC_P0_HI, (uint32_t)(st.P >> 32));
      { const uint32_t rl = pe_ctl_ld(pb, PEC_RELAX), g26 = rdlane((uint32_t)v, 26), g27 = rdlane((uint32_t)v, 27);   // (the sizes of the two regions before this one)
        pe_ctl_st(pb, PEC_PREVOUT, g26); pe_ctl_st(pb, PEC_DEPTH, rl);
        pe_ctl_st(pb, PEC_LAG, rl == 2u ? g26 + g27 : rl == 1u ? g26 : 0u); }
    }
    if (ok_ == 0u) {
      if (!arrived) GANG_STAT(gc, 32, 1); else if ((rdlane((uint32_t)v, 25) & 1u) == 0u) GANG_STAT(gc, 33, 1); else { const PeStream st = pe_st_load(pbs); if (st.b != pe_ctl_ld(pb, PEC_MYENTRY)) GANG_STAT(gc, 34, 1); else GANG_STAT(gc, 36, 1); }
      if (arrived && lane == 0u) gang_st64(gc, GC_STOP, ((uint64_t)epoch << 32) | (uint64_t)kseq);   // (the same word, if the resolve before has written it)
      pe_ctl_st(pb, PEC_CONT, 0u);
    }
    pe_ctl_st(pb, PEC_PLAN, ok_ != 0u ? 0u : 2u);
  };
  // ================= the stream's way through the region: walk, details, resolve, execute =================
  auto consume = [&]() {
    rseq++;
    // ---- the walk (wave 0) and, behind it, the details (the other waves): the stream's states in order, LIST[k] = bit | kind << 15
    // of the state command k starts from.  Wave 0 follows the stream eight commands a hop (NEXT8 knows the way wherever the next
    // eight records are ordinary ones: everywhere but at the region's end) and publishes every anchor as it finds it; then
    // command by command up to the first record that is no way on (END: the run leaves the region; BYHAND: the closure ran
    // out of room -- the next region starts there).  Wave w + 1 takes batch w -- commands 64 w .. 64 w + 63 -- as soon as the nine
    // anchors that span it are there (or the walk is over): lane = command, the state it starts from by following the anchor's
    // records, then its fields parsed once more, its distance from the state after.  The fields stay in the lane's registers:
    // the resolve below is by the same wave.
    const uint32_t bw = (me + GW - 1u) & (GW - 1u);  // this wave's batch (wave 0, which walks, gets the last one)
    uint32_t dr0 = 0, dr1 = 0, dr2 = 0, dr3 = 0;
    if (me == 0) {
      if (REMOTE) GANG_STAT(gc, 28, __builtin_amdgcn_s_memtime() - gs_arr);   // arrival .. the walk's start
      __builtin_amdgcn_s_setprio(3);  // (the walk is the one chain everybody waits for: first in line on its SIMD)
      uint32_t id = PE_RANKS, na = 0;
      uint32_t id_hand = PEN_NONE;   // (two engines) the first of the states the walk added itself: NEXT8 does not know them
      if (PIPE) {
        // The tables were built before the stream's entry into the region was known: the state it enters in -- a command starts
        // at bit `le` -- is evaluated here, and the states it leads to, until one of them is a path state (as a rule the first
        // or the second).  They join the closure behind the ones the records put there.
        uint32_t slot = wn, desc = le | 0x8000u;
        id = PE_RANKS + slot; id_hand = id;
        bool long_run = false;
        if (REMOTE && le + 64u <= c.L) {
          // (a gang: a first command with a literal run that wants regions of its own is not looked at any closer -- evaluated here, its run
          // would be followed one code word after the other by this wave alone; nothing listed, and the resolve says why: see PEC_DECLINE)
          uint32_t lo_, hi_;
          pe_bits64(pb, le, lo_, hi_);
          const ScHead h_ = sc_head(lo_, hi_, c.cmd_tree, c.lut_vgpr);
          long_run = rfl(h_.insert) >= PE_RUN_MIN;
        }
        bool seeded = false;
        if (REMOTE && !long_run) {   // (a gang) the state the stream enters in is one of the window's entry seeds: its record is there, and NEXT8 knows it
          const uint32_t sn_ = pe_ctl_ld(pb, PEC_SEEDN), off_ = le - pe_ctl_ld(pb, PEC_SEEDLO);
          if (off_ < sn_) { id = PE_RANKS + 1u + off_; id_hand = PEN_NONE; seeded = true; }
#ifdef BROTLI_AMD_SEED_DEBUG
          if (lane == 0 && kseq < 400u) printf("seed: region %u role %u le %u lo %u n %u hit %d L %u\n", kseq, role, le, pe_ctl_ld(pb, PEC_SEEDLO), sn_, (int)seeded, c.L);
#endif
        }
        if (seeded) { }
        else if (slot >= PE_WCAP || long_run) id = PEN_NONE;   // (no room for the entry's state: nothing listed, the checked loop's)
        else for (uint32_t tries = 0;; tries++) {
          if (lane == 0) lds_st16(pb + PE_WST + (slot << 1), desc);
          if (tries >= PE_PIPE_HAND) { if (lane == 0) lds_st16(pb + PE_NEXT + ((PE_RANKS + slot) << 1), PEN_BYHAND); break; }   // (the region ends in front of this state)
          const PeParse pr = pe_eval<false, false>(c, desc & 0x7FFFu, desc >> 15, true);
          const uint32_t cd = rfl(pr.code), nx = rfl(pr.next);
          const bool more = cd == 1u && slot + 1u < PE_WCAP;
          if (lane == 0) lds_st16(pb + PE_NEXT + ((PE_RANKS + slot) << 1), cd == 0u ? nx : more ? PE_RANKS + slot + 1u : cd == 1u ? (uint32_t)PEN_BYHAND : (uint32_t)PEN_END);
          if (!more) break;
          desc = nx; slot++;
        }
        lds_sync();
      }
#if !PE_CFG_PIPE && !defined(BROTLI_AMD_PE_NO_WALK_ASM)
      bool walk_on = true;
      if (REMOTE) {
        // (a gang: the anchors up to the first one that stands on a state NEXT8 knows -- as a rule the first -- by the records, eight hops each)
        while (id >= id_hand && id < PEN_FIRST_SPECIAL) {
          uint32_t n8 = id;
          for (uint32_t h = 0; h < PE_JUMP; h++) n8 = n8 < PEN_FIRST_SPECIAL ? rfl(lds_ld16(pb + PE_NEXT + (n8 << 1))) : (uint32_t)PEN_NONE;
          if (n8 >= PEN_FIRST_SPECIAL || na >= (PE_CMDS - 64u) / PE_JUMP) { walk_on = false; break; }
          if (lane == 0) {
            *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_ANCH + (na << 2)]) = id;
            *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_NAPUB]) = na + 1u;
          }
          na++; id = n8;
        }
        if (id >= PEN_FIRST_SPECIAL) walk_on = false;
      }
      if (REMOTE) { id = rfl(id); na = rfl(na); }   // (uniform, and in scalar registers for what follows)
      if (REMOTE) GANG_STAT(gc, 29, __builtin_amdgcn_s_memtime() - gs_arr);   // .. the entry's states and the first anchor
      if (walk_on) {
        // The anchors by hand: one dependent LDS read an anchor is all the chain asks for, and the compiled loop wrapped it in
        // thirty-five instructions (the lane's own execution mask, the counter in a vector register): 330 clocks an anchor.
        // Here lane 0 alone: the next anchor's read is on its way before this one is stored and published.
        uint32_t vr, va, vt; uint64_t sv; uint32_t n8s, ts, aa = pb + PE_ANCH + (na << 2);
        const uint32_t n8base = pb + PE_N8, pubaddr = pb + PE_CTL + 4u * PEC_NAPUB, cap = (PE_CMDS - 64u) / PE_JUMP;
        asm volatile(
          "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\t"
          "s_lshl_b32 %[ts], %[id], 1\n\ts_add_u32 %[ts], %[ts], %[n8base]\n\tv_mov_b32 %[va], %[ts]\n\tds_read_u16 %[vr], %[va]\n"
          ".Lpe_walk_%=:\n\t"
          "s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %[n8s], %[vr]\n\t"
          "s_cmp_ge_u32 %[n8s], 0xfff0\n\ts_cbranch_scc1 .Lpe_walk_done_%=\n\t"
          "s_cmp_ge_u32 %[na], %[cap]\n\ts_cbranch_scc1 .Lpe_walk_done_%=\n\t"
          "s_lshl_b32 %[ts], %[n8s], 1\n\ts_add_u32 %[ts], %[ts], %[n8base]\n\tv_mov_b32 %[va], %[ts]\n\tds_read_u16 %[vr], %[va]\n\t"
          "v_mov_b32 %[vt], %[id]\n\tv_mov_b32 %[va], %[aa]\n\tds_write_b32 %[va], %[vt]\n\t"
          "s_add_u32 %[na], %[na], 1\n\ts_add_u32 %[aa], %[aa], 4\n\t"
          "v_mov_b32 %[vt], %[na]\n\tv_mov_b32 %[va], %[pub]\n\tds_write_b32 %[va], %[vt]\n\t"
          "s_mov_b32 %[id], %[n8s]\n\ts_branch .Lpe_walk_%=\n"
          ".Lpe_walk_done_%=:\n\ts_mov_b64 exec, %[sv]"
          : [id] "+s"(id), [na] "+s"(na), [aa] "+s"(aa), [vr] "=&v"(vr), [va] "=&v"(va), [vt] "=&v"(vt), [sv] "=&s"(sv), [n8s] "=&s"(n8s), [ts] "=&s"(ts)
          : [n8base] "s"(n8base), [pub] "s"(pubaddr), [cap] "s"(cap) : "scc", "memory");
      }
#else
      for (;;) {
        uint32_t n8;
        if (PIPE && id >= id_hand && id < PEN_FIRST_SPECIAL) {   // eight records on from a state NEXT8 has not seen (the walk's own): by the records
          n8 = id;
          for (uint32_t h = 0; h < PE_JUMP; h++) n8 = n8 < PEN_FIRST_SPECIAL ? rfl(lds_ld16(pb + PE_NEXT + (n8 << 1))) : (uint32_t)PEN_NONE;
        } else n8 = id < PEN_FIRST_SPECIAL ? rfl(lds_ld16(pb + PE_N8 + (id << 1))) : (uint32_t)PEN_NONE;
        if (n8 >= PEN_FIRST_SPECIAL || na >= (PE_CMDS - 64u) / PE_JUMP) break;
        if (lane == 0) {
          *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_ANCH + (na << 2)]) = id;
          *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_NAPUB]) = na + 1u;
        }
        na++; id = n8;
      }
#endif
      // ... then the commands behind the last anchor, up to the first record that is no way on: lane j follows the records j
      // commands on (the lanes side by side: a dozen dependent reads for all of them, where one command after the other by the
      // wave as a whole cost a third of the walk), the list's entries are theirs
      if (!PIPE) PE_PROF(15);   // (one engine: the anchors)
      if (REMOTE) GANG_STAT(gc, 38, __builtin_amdgcn_s_memtime() - gs_arr);   // .. the anchors
      uint32_t m = PE_JUMP * na, desc;
      if (PIPE && id >= PEN_FIRST_SPECIAL) { desc = le | 0x8000u; if (lane == 0) lds_st16(pb + PE_LIST, desc); }   // (no room for the entry's state: nothing listed -- the list's closing entry says where the stream stands)
      else for (;;) {
        uint32_t sv = id;
        for (uint32_t h = 0; h < 63u; h++) {
          const uint32_t nxv = lds_ld16(pb + PE_NEXT + ((sv < PE_STATES ? sv : PE_STATES) << 1));   // (a record that is no way on reads the sentinel: NONE)
          if (lane > h) sv = nxv;
          if (rdlane(sv, h + 1u) >= PEN_FIRST_SPECIAL) break;
        }
        const uint32_t svc = sv < PE_STATES ? sv : PE_STATES;
        const uint32_t nxv = lds_ld16(pb + PE_NEXT + (svc << 1));
        const uint32_t dv = lds_ld16(pb + (svc < PE_RANKS ? PE_POR + (svc << 1) : svc < PE_STATES ? PE_WST + ((svc - PE_RANKS) << 1) : PE_CTL + 4u * PEC_SCRATCH));
        const uint64_t lm = __ballot(sv < PEN_FIRST_SPECIAL && nxv < PEN_FIRST_SPECIAL && m + lane < PE_CMDS);
        const uint32_t J = ~lm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~lm);   // commands listed here: lanes 0 .. J - 1; lane J's state closes the list
        if (lane <= J && lane < 64u && sv < PEN_FIRST_SPECIAL) lds_st16(pb + PE_LIST + ((m + lane) << 1), dv);
        if (J < 64u) { m += J; desc = rdlane(dv, J); id = rdlane(sv, J); break; }
        m += 64u; id = rdlane(nxv, 63);
      }
      if (!PIPE) PE_PROF(16);   // (one engine: the commands behind the last anchor)
      // the last command needs its distance: 64 bits at the closing state
      if (m != 0u && (desc >> 15) == 0u && (desc & 0x7FFFu) + 64u > c.L) m--;
#ifdef BROTLI_AMD_PE_DEBUG
      if (PIPE && blockIdx.x == 0 && lane == 0 && kseq >= 33u && kseq <= 36u) {
        printf("   walk of region %u: le %u, id_hand %u, anchors %u, listed %u, closing state %x (id %u), L %u Lp %u\n", kseq, le, id_hand, na, m, desc, id, c.L, c.Lp);
        for (uint32_t q = 0; q < 4u; q++) printf("     hand state %u: desc %x next %u\n", q, lds_ld16(pb + PE_WST + ((wn + q) << 1)), lds_ld16(pb + PE_NEXT + ((PE_RANKS + wn + q) << 1)));
        for (uint32_t q = 0; q < (m < 12u ? m + 1u : 12u); q++) printf("     list %u: %x\n", q, lds_ld16(pb + PE_LIST + (q << 1)));
      }
#endif
      if (REMOTE && m != 0u) {
        // (a gang) where the stream goes on if every command listed goes through -- the next region's engine starts its walk from there while
        // this region is resolved: the first bit of the command the list closes with (behind the distance code, if it starts with one)
        lds_sync();
        uint32_t dsc;
        if (m >= PE_JUMP * na) dsc = rfl(lds_ld16(pb + PE_LIST + (m << 1)));
        else {   // (the list's entries up to the last anchor are the details' to write: by the anchor and the records, as they do it)
          uint32_t st_ = rfl(*reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_ANCH + ((m >> PE_JUMP_LOG) << 2)]));
          for (uint32_t h = 0; h < (m & (PE_JUMP - 1u)); h++) st_ = rfl(lds_ld16(pb + PE_NEXT + (st_ << 1)));
          dsc = rfl(st_ < PE_RANKS ? lds_ld16(pb + PE_POR + (st_ << 1)) : lds_ld16(pb + PE_WST + ((st_ - PE_RANKS) << 1)));
        }
        uint32_t nbit = dsc & 0x7FFFu;
        if ((dsc >> 15) == 0u) {
          uint32_t lo_, hi_;
          pe_bits64(pb, nbit, lo_, hi_);
          const ScDist d_ = sc_dist(lo_, hi_, c.dtree, c.postfix_bits, c.num_direct);
          nbit += rfl(d_.bits);
        }
        {   // (regions that were halved take twice the bits again from the next one on where this one's closure is small: a new plan in front of the entry)
          const uint32_t shsc_ = pe_ctl_ld(pb, PEC_MYSHIFT), sh_ = shsc_ & 3u;
          if (sh_ != 0u && wn < PE_WCAP / 8u && kseq + 1u < GC_MAX_REGIONS) {
            const uint64_t cur = gang_ld64(gc, GC_PLAN);
            if ((uint32_t)(cur >> 48) == pe_ctl_ld(pb, PEC_MYGEN)) {   // (nobody has changed it since this engine's window was laid out)
              if (lane == 0u) gang_st64(gc, GC_PLAN, ((uint64_t)((pe_ctl_ld(pb, PEC_MYGEN) + 1u) & 0xFFFFu) << 48) | ((uint64_t)(shsc_ - 1u) << 44) | ((uint64_t)(kseq + 1u) << 32) | (uint64_t)((pe_ctl_ld(pb, PEC_LBDW) << 5) + nbit));
              gang_drain();
            }
          }
        }
        if (lane == 0u) gang_st64(gc, GC_ENTRY, (uint64_t)((pe_ctl_ld(pb, PEC_LBDW) << 5) + nbit) | ((uint64_t)((epoch << 12) | (kseq + 1u)) << 32));
        pe_ctl_st(pb, PEC_MYNEXT, (pe_ctl_ld(pb, PEC_LBDW) << 5) + nbit);
      }
      pe_ctl_st(pb, PEC_M, m); pe_ctl_st(pb, PEC_NA, na);
      lds_sync();
      pe_ctl_st(pb, PEC_WDONE, 1u);
      __builtin_amdgcn_s_setprio(BROTLI_AMD_DECODER_PRIO);  // (back to the decoding wave's own)
      PE_COUNT(26, m); PE_COUNT(27, na);
    }
    PE_PROF(6);
    if (REMOTE && me == 0) GANG_STAT(gc, 24, __builtin_amdgcn_s_memtime() - gs_arr);   // arrival .. walk done
    if (REMOTE && me == 0) GT(1);
    {
      const uint32_t k0 = bw << 6;
      uint32_t na_k, m_k;
      for (;;) {
        const uint32_t done = pe_ctl_ld(pb, PEC_WDONE), pub = pe_ctl_ld(pb, PEC_NAPUB);
        if (done != 0u) { lds_sync(); na_k = pe_ctl_ld(pb, PEC_NA); m_k = pe_ctl_ld(pb, PEC_M); break; }
        if (pub >= (k0 >> PE_JUMP_LOG) + 64u / PE_JUMP + 1u) { na_k = pub; m_k = PE_CMDS + 64u; break; }  // (the anchors that span it and one more: the batch is whole whatever comes behind)
        __builtin_amdgcn_s_sleep(BROTLI_AMD_PE_POLL_SLEEP);
      }
      lds_sync();
      if (k0 <= m_k) {
        // the state command kk starts from: behind an anchor by the records, else what the walk listed
        auto state_of = [&](const uint32_t kk) -> uint32_t {
          if (kk >= PE_JUMP * na_k) return lds_ld16(pb + PE_LIST + ((kk < PE_CMDS + 8u ? kk : 0u) << 1));
          uint32_t st = *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_ANCH + ((kk >> PE_JUMP_LOG) << 2)]);
          for (uint32_t h = 0; h < (kk & (PE_JUMP - 1u)); h++) st = lds_ld16(pb + PE_NEXT + (st << 1));
          return st < PE_RANKS ? lds_ld16(pb + PE_POR + (st << 1)) : lds_ld16(pb + PE_WST + ((st - PE_RANKS) << 1));
        };
        const uint32_t k = k0 + lane;
        const bool on = k < m_k;
        const uint32_t s0 = state_of(k <= m_k ? k : 0u);
        const uint32_t s_last = state_of(k0 + 64u <= m_k ? k0 + 64u : 0u);
        uint32_t s1 = bperm(((lane + 1u) & 63u) << 2, s0);
        s1 = lane == 63u ? s_last : s1;
        if (k <= m_k && k < PE_JUMP * na_k) lds_st16(pb + PE_LIST + (k << 1), s0);  // (the resolve reads where the stream goes on out of the list)
        if (k0 < m_k) {
          const PeParse pr = pe_eval<false, false>(c, s0 & 0x7FFFu, s0 >> 15, on);
          uint32_t kind = SCK_IMPLICIT, val = 0;
          if (__ballot(on && (s1 >> 15) == 0u) != 0ull) {
            uint32_t lo, hi;
            pe_bits64(pb, on ? (s1 & 0x7FFFu) : 0u, lo, hi);
            const ScDist d = sc_dist(lo, hi, c.dtree, c.postfix_bits, c.num_direct);
            if ((s1 >> 15) == 0u) { kind = d.kind; val = d.val; }
          }
          // a command the fields do not hold goes to the checked loop (bit 30 of w0: the resolve stops in front of it)
          const bool odd = pr.u > 255u || pr.insert >= 0x10000u || val >= (1u << 30) || pr.code >= 2u;
          if (on) {
            dr0 = pr.x | ((pr.u & 255u) << 15) | (odd ? 1u << 30 : 0u); dr1 = (pr.insert & 0xFFFFu) | (pr.ry << 16);
            dr2 = pr.copy; dr3 = (kind << 30) | (val & 0x3FFFFFFFu);
          }
        }
      }
    }
    PE_BAR();
    RG_STAMP(0);   // walk + details done
    if (REMOTE && me == 0) GANG_STAT(gc, 25, __builtin_amdgcn_s_memtime() - gs_arr);   // .. details done
    if (REMOTE && me == 0) GT(2);
    const uint32_t m = pe_ctl_ld(pb, PEC_M);
    PE_PROF(7);
    // ---- resolve: wave w takes batch w (64 commands), all batches side by side.  What one batch needs from the batches in front
    // of it -- the sums (literals, commands, distances, output bytes) and the distance ring (decode.rs:2017-2049) -- goes
    // through LDS: every batch resolves its ring against an UNKNOWN ring at its start (a distance is a constant, or one of
    // the four entries it started with plus a small delta), leaves its sums and the ring it ends with in that form, and
    // after a barrier every wave puts the batches in front of it together.  Then every limit the reference checks, as in
    // the scan engine; the first command that needs the checked loop ends the engine's part in front of it.
    const uint32_t nb = (m + 63u) >> 6;
    uint32_t ks = 0;       // the pass's first command (PE_DICT: the commands in front of it went out in the passes before)
pe_pass:
    uint32_t my_exec = 0;  // commands of this wave's batch that are executed (counted from the batch's first: those below ks are not)
    {
      // (a gang: the stream's state is the region before's resolve's to send, and what this resolve does in front of its first barrier -- the sums, the ring
      // against an unknown ring -- does not ask for it: wave 0 waits for it behind that, see full_arrival)
      PeStream st{};
      if (!REMOTE) st = pe_st_load(pbs);
      const bool mine = bw < nb;
      const uint32_t k0 = bw << 6;
      const uint32_t K = mine ? (m - k0 < 64u ? m - k0 : 64u) : 0u;
      const bool active = lane < K && k0 + lane >= ks;
      const uint32_t ra = pb + PE_REC + ((mine ? k0 + (active ? lane : 0u) : 0u) << 4);
      const uint32_t r0 = dr0, r1 = dr1, r2 = dr2, r3 = dr3;  // (the details above: this wave's batch, in its registers)
      const uint32_t ins = active ? r1 & 0xFFFFu : 0u, copy = active ? r2 : 0u;
      const uint32_t kind = active ? r3 >> 30 : (uint32_t)SCK_NONE, val = r3 & 0x3FFFFFFFu;
      const bool odd = ((r0 >> 30) & 1u) != 0u;
      const uint32_t isdist = (kind == SCK_EXPLICIT || kind == SCK_SHORT) ? 1u : 0u;
      const uint32_t lit_incl = sc_scan(ins);
      const uint32_t s1 = sc_scan((active ? 1u : 0u) | (isdist << 16));
      // (a sum of 64 copy lengths stays below 2^31: lengths of 2^24 and more are `odd` below)
      const bool big = copy >= (1u << 24);
      const uint32_t s2 = sc_scan(ins + (big ? 0u : copy));
      const uint32_t out_excl = s2 - (ins + (big ? 0u : copy));
      // the ring, against an unknown ring at the batch's start: (dtag, dval) = entry dtag of that ring plus dval, or (4, the distance)
      const bool need = kind == SCK_SHORT || kind == SCK_IMPLICIT;
      const uint32_t code = kind == SCK_SHORT ? val : 0u;
      const uint32_t rs = pb + PE_RS + (bw << 6);
      uint64_t pmk; uint32_t dtag; int32_t dval; uint32_t n_all, perm;
      // (`pushes`: an explicit distance, or a ring code other than "the last one" -- but not the distance of a dictionary word,
      // decode.rs:2643-2644: PE_DICT resolves a second time once it knows which commands those are)
      auto ring_resolve = [&](const bool pushes) {
        pmk = __ballot(pushes);
        dtag = kind == SCK_EXPLICIT ? 4u : 0u; dval = kind == SCK_EXPLICIT ? (int32_t)val : 0;
        const uint32_t npush = __builtin_amdgcn_mbcnt_hi((uint32_t)(pmk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmk, 0u));  // pushes in front of this lane
        n_all = (uint32_t)__popcll(pmk);
        perm = (uint32_t)__builtin_amdgcn_ds_permute((int)((pushes ? npush : n_all + lane - npush) << 2), (int)lane);
        if (__ballot(need) != 0ull) {
          const uint32_t back = code == 0u ? 0u : 3u - ((0xaaafff1bu >> (code << 1)) & 3u);
          const bool from_carry = npush <= back;
          const uint32_t ci = back - npush;  // (meaningful when from_carry)
          const uint32_t src = bperm(((npush - 1u - back) & 63u) << 2, perm);
          uint32_t resolved = need ? 0u : 1u;
          const int32_t mag = (int32_t)((0xfa5fa500u >> (code << 1)) & 3u);
          while (__ballot(resolved == 0u) != 0ull) {
            const uint32_t stg = bperm(src << 2, dtag);
            const int32_t sv = (int32_t)bperm(src << 2, (uint32_t)dval);
            const uint32_t sr = bperm(src << 2, resolved);
            const bool can = resolved == 0u && (from_carry || sr != 0u);
            const uint32_t bt = from_carry ? ci : stg;
            const int32_t bv = from_carry ? 0 : sv;
            const int32_t nv = code == 0u ? bv : (code & 1u) ? bv + mag : bv - mag;  // (a result <= 0 is invalid: seen once the ring is known)
            dtag = can ? bt : dtag; dval = can ? nv : dval; resolved = can ? 1u : resolved;
          }
        }
        // the ring the batch ends with (all its pushes: a batch that stops short is the last one that counts)
        if (mine) {
          const uint32_t tperm = bperm(perm << 2, dtag), vperm = bperm(perm << 2, (uint32_t)dval);  // lane r: the r-th push
          _Pragma("unroll") for (uint32_t r = 0; r < 4u; r++) {
            const uint32_t tg = n_all > r ? rdlane(tperm, (n_all - 1u - r) & 63u) : r - n_all;
            const uint32_t vl = n_all > r ? rdlane(vperm, (n_all - 1u - r) & 63u) : 0u;
            if (lane == 0) { lds_st32(rs + 16u + 8u * r, tg); lds_st32(rs + 20u + 8u * r, vl); }
          }
        }
      };
      const bool pushes0 = kind == SCK_EXPLICIT || (kind == SCK_SHORT && val != 0u);
      ring_resolve(pushes0);
      // the batch's sums
      if (mine && lane == 0) { lds_st32(rs, rdlane(lit_incl, 63)); lds_st32(rs + 4u, rdlane(s1, 63)); lds_st32(rs + 8u, rdlane(s2, 63)); }
      if (PE_DICT) {
        // may any command of the pass be a word of the static dictionary?  An explicit distance beyond what the stream had put out
        // when the pass began (or beyond the window) is the only kind that can be (decode.rs:2583-2593): where there is none -- the
        // rule -- the pass is as it was
        const uint32_t reach = st.P < (uint64_t)(uint32_t)st.max_backward ? (uint32_t)st.P : (uint32_t)st.max_backward;
        if (__ballot(active && kind == SCK_EXPLICIT && val > reach) != 0ull && lane == 0) *reinterpret_cast<lds_vu32*>(&g_smem[pb + PE_CTL + 4u * PEC_DCAND]) = 1u;
      }
      if (T == 0u) { lds_st32(pb + PE_CTL + 4u * PEC_KP, m); lds_st32(pb + PE_CTL + 4u * PEC_BIGNEXT, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DICTK, 0xFFFFFFFFu); lds_st32(pb + PE_CTL + 4u * PEC_WNEXT, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DEPCHG, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DEPLV0, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DEPLV1, 0u); lds_st32(pb + PE_CTL + 4u * PEC_DEPDEEP, 0u); }   // (... and the execute's items are handed out from the first)
      if (REMOTE && me == 0) { full_arrival(true); GT(3); }
      PE_BAR();
      if (REMOTE) {
        if (pe_ctl_ld(pb, PEC_PLAN) == 2u) return;
        P0 = (uint64_t)pe_ctl_ld(pb, PEC_P0_LO) | ((uint64_t)pe_ctl_ld(pb, PEC_P0_HI) << 32);
        st = pe_st_load(pbs);
      }
      // what lies in front of this batch
      uint32_t c_lit = 0, c_cmd = 0, c_dst = 0, c_out = 0;
      int32_t d0 = st.d0, d1 = st.d1, d2 = st.d2, d3 = st.d3;
#define PE_RING_AT(tg_, vl_) ((tg_) == 4u ? (int32_t)(vl_) : ((tg_) == 0u ? o0 : (tg_) == 1u ? o1 : (tg_) == 2u ? o2 : o3) + (int32_t)(vl_))
      for (uint32_t j = 0; j < bw && mine; j++) {
        const uint32_t rj = pb + PE_RS + (j << 6);
        const uint32_t w = lds_ld32(rj + (lane < 12u ? lane << 2 : 0u));   // (one read: lane k word k)
        c_lit += rdlane(w, 0); const uint32_t t1 = rdlane(w, 1); c_cmd += t1 & 0xFFFFu; c_dst += t1 >> 16; c_out += rdlane(w, 2);
        const int32_t o0 = d0, o1 = d1, o2 = d2, o3 = d3;
        d0 = PE_RING_AT(rdlane(w, 4), rdlane(w, 5)); d1 = PE_RING_AT(rdlane(w, 6), rdlane(w, 7));
        d2 = PE_RING_AT(rdlane(w, 8), rdlane(w, 9)); d3 = PE_RING_AT(rdlane(w, 10), rdlane(w, 11));
      }
      // (PE_DICT) Words of the static dictionary INSIDE the pass.  A word's output is not its copy length long (transform.rs:737-795),
      // and while the window is not full yet its number depends on where it stands (decode.rs:2583-2603: distance - max_distance - 1),
      // i.e. on the lengths of the words before it.  Round 4 ended a pass at every word (a resolve and an execute per word: three
      // quarters of a text stream's time at -q 5).  Now a fixed point over the whole region: every lane classifies its command with the
      // words in front of it as the round before left them (`dex`: what they add to, or take from, the copy lengths' sum), the
      // lengths' differences are summed up over the block, and when a round changes no lane's difference the classification is the
      // stream's.  A word's length follows from its copy length and transform alone, and a shift of a few bytes rarely moves a word to
      // another transform: two rounds as a rule, a barrier each.  What does not settle in eight falls back to a pass per word.
      bool isw = false, plainw = false;   // the copy is a dictionary reference; ... one that goes out inside this pass
      uint32_t wdesc = 0, wdelta = 0, dex = 0;   // the word (offset | transform << 17 | length << 24); its length less the copy's; that, summed over the commands in front of this one
      if (PE_DICT && pe_ctl_ld(pb, PEC_DCAND) != 0u) {
        auto classify = [&](const bool plain_ok) {
          const uint64_t pk = st.P + (uint64_t)(c_out + out_excl + ins + dex);
          const uint32_t maxd = pk < (uint64_t)(uint32_t)st.max_backward ? (uint32_t)pk : (uint32_t)st.max_backward;
          isw = (bool)((uint32_t)active & (uint32_t)!odd & (uint32_t)!big & (uint32_t)(kind == SCK_EXPLICIT) & (uint32_t)(val > maxd));
          plainw = false; wdesc = 0; wdelta = 0;
          if (plain_ok && isw && copy >= 4u && copy <= 24u) {
            const uint32_t shift = kDictSizeBitsByLength[copy], id = val - maxd - 1u, tix = id >> shift;
            if (tix < (uint32_t)BROTLI_NUM_TRANSFORMS) {
              const WordShape w = word_shape(copy, tix);
              if (w.total != 0u) { plainw = true; wdelta = w.total - copy; wdesc = (kDictOffsetsByLength[copy] + (id & mask_bits(shift)) * copy) | (tix << 17) | (copy << 24); }
            }
          }
        };
        uint32_t prev = 0; bool settled = false;
        for (uint32_t it = 0; it < 8u; it++) {
          classify(true);
          const uint32_t dincl = sc_scan(wdelta);
          const bool chg = __ballot(wdelta != prev) != 0ull;
          prev = wdelta;
          const uint32_t slot = 52u + 4u * (it & 1u);
          if (lane == 0) lds_st32(rs + slot, (rdlane(dincl, 63) << 1) | (chg ? 1u : 0u));   // (every wave: one without a batch says 0)
          PE_BAR();
          const uint32_t wv = lane < GW ? lds_ld32(pb + PE_RS + (lane << 6) + slot) : 0u;
          const bool anychg = __ballot((wv & 1u) != 0u) != 0ull;
          const uint32_t sincl = sc_scan((uint32_t)((int32_t)wv >> 1));
          dex = (bw == 0u ? 0u : rdlane(sincl, (bw - 1u) & 63u)) + dincl - wdelta;
          if (!anychg) { settled = true; break; }   // (this round's differences were the round before's: so is what they sum up to)
        }
        if (!settled) { dex = 0; classify(false); }   // (up to the first word nothing has moved: the pass ends there, as round 4's did)
        // ... and the ring once more: a dictionary word's distance is not pushed
        if (__ballot(isw) != 0ull || true) {   // (uniform over the block: every wave resolves again and meets the others at the barrier)
          ring_resolve(pushes0 && !isw);
          PE_BAR();
          d0 = st.d0; d1 = st.d1; d2 = st.d2; d3 = st.d3;
          for (uint32_t j = 0; j < bw && mine; j++) {
            const uint32_t rj = pb + PE_RS + (j << 6);
            const uint32_t w = lds_ld32(rj + (lane < 12u ? lane << 2 : 0u));
            const int32_t o0 = d0, o1 = d1, o2 = d2, o3 = d3;
            d0 = PE_RING_AT(rdlane(w, 4), rdlane(w, 5)); d1 = PE_RING_AT(rdlane(w, 6), rdlane(w, 7));
            d2 = PE_RING_AT(rdlane(w, 8), rdlane(w, 9)); d3 = PE_RING_AT(rdlane(w, 10), rdlane(w, 11));
          }
        }
      }
#undef PE_RING_AT
      const int32_t dist = dtag == 4u ? dval : (dtag == 0u ? d0 : dtag == 1u ? d1 : dtag == 2u ? d2 : d3) + dval;
      const uint32_t lit_a = c_lit + lit_incl, cmd_a = c_cmd + (s1 & 0xFFFFu), dst_a = c_dst + (s1 >> 16);
      const uint32_t dcum = dex + wdelta;   // (the words' differences up to and with this command's)
      const uint64_t out_a = (uint64_t)(c_out + s2 + dcum);
      bool ok = !odd && !big && lit_a <= st.bl0 && cmd_a <= st.bl1 && dst_a <= st.bl2 && out_a < (uint64_t)st.quota;
      const uint64_t rel = (uint64_t)(c_out + out_excl + dex);   // where the command's output starts, from the region's
      bool dictc = false;   // (PE_DICT) the copy is a dictionary reference that does NOT go out inside the pass, and the command's literals clear every limit
      bool dref = false;    // (lean form) ... a dictionary reference at all
      {
        // max distance at the copy (decode.rs:2583-2589); beyond it the distance names a dictionary word
        const uint64_t pk = st.P + rel + ins;
        const int32_t maxd = pk < (uint64_t)(uint32_t)st.max_backward ? (int32_t)pk : st.max_backward;
        if (PE_DICT) dictc = (bool)((uint32_t)ok & (uint32_t)(kind == SCK_EXPLICIT) & (uint32_t)(dist > maxd) & (uint32_t)!plainw);   // (`ok` so far: an active lane, its counts and its output -- the copy's length for the word's -- inside every limit)
        if (!PE_DICT && !PIPE2) dref = (bool)((uint32_t)ok & (uint32_t)(kind == SCK_EXPLICIT) & (uint32_t)(dist > maxd));
        ok = ok && (kind == SCK_NONE || plainw || (dist > 0 && dist <= maxd));
      }
      const uint64_t stopmask = __ballot(active && !ok);
      uint32_t kpb = stopmask ? (uint32_t)__builtin_ctzll(stopmask) : K;
      const uint64_t dictmask = PE_DICT ? __ballot(dictc) : 0ull;
      const uint64_t drefmask = (!PE_DICT && !PIPE2) ? __ballot(dref) : 0ull;
      if (PE_DICT && stopmask != 0ull && ((dictmask >> kpb) & 1ull) != 0ull) kpb++;   // (the pass ends BEHIND such a command's literals)
      if (mine && stopmask != 0ull && lane == 0) pe_atomic_min(pb + PE_CTL + 4u * PEC_KP, k0 + kpb);
      // a copy whose source reaches into the region's own output is done afterwards (bit 31 of w0); long items get a wave
      const uint32_t copy_x = (dictc || plainw) ? 0u : copy;   // (a dictionary word is no LZ77 copy: wave 0's behind the pass, or -- inside it -- an item of its own list)
      // (a gang: a copy that reads the region BEFORE's output -- the last `lag` bytes in front of this region's -- waits with the copies that read
      // this region's own: that region's engine may still be writing them.  What lies in front of that is there: see the execute's two waits.)
      const uint32_t dep = (copy_x != 0u && rel + ins + copy_x + (REMOTE ? (uint64_t)pe_ctl_ld(pb, PEC_LAG) : 0ull) > (uint64_t)(uint32_t)dist) ? 1u : 0u;
      uint32_t uu = (r0 >> 15) & 255u; uu = uu < ins ? uu : ins;
      const bool bigc = (copy_x > PE_LANE_COPY && dep == 0u) || ins - uu > PE_LANE_LITS;
      const uint64_t dmk = __ballot(lane < kpb && dep != 0u), bmk = __ballot(lane < kpb && bigc), wmk = PE_DICT ? __ballot(lane < kpb && plainw) : 0ull;
      if (mine && lane == 0) { lds_st32(rs + 48u, (uint32_t)__popcll(dmk) | ((uint32_t)__popcll(bmk) << 16)); if (PE_DICT) lds_st32(rs + 12u, (uint32_t)__popcll(wmk)); }
      PE_BAR();
      const uint32_t kp_total = pe_ctl_ld(pb, PEC_KP);
      my_exec = !mine || kp_total <= k0 ? 0u : (kp_total - k0 < K ? kp_total - k0 : K);
      if (mine && my_exec != 0u) {
        // (every batch in front of one that executes anything went through whole: its counts are its lists' lengths)
        uint32_t c_dep = 0, c_big = 0, c_word = 0;
        {
          const uint32_t w = lane < bw ? lds_ld32(pb + PE_RS + (lane << 6) + 48u) : 0u;
          const uint32_t sum = sc_scan(w);  // (two 16-bit sums side by side: at most 1024 each)
          const uint32_t tot = rdlane(sum, 63);
          c_dep = tot & 0xFFFFu; c_big = tot >> 16;
          if (PE_DICT && pe_ctl_ld(pb, PEC_DCAND) != 0u) c_word = rdlane(sc_scan(lane < bw ? lds_ld32(pb + PE_RS + (lane << 6) + 12u) : 0u), 63);
        }
        const uint64_t wm2 = wmk & ((my_exec >= 64u) ? ~0ull : ((1ull << my_exec) - 1ull));
        const uint64_t dm2 = dmk & ((my_exec >= 64u) ? ~0ull : ((1ull << my_exec) - 1ull)), bm2 = bmk & ((my_exec >= 64u) ? ~0ull : ((1ull << my_exec) - 1ull));
        if (lane < my_exec && active) {
          if (dep != 0u) lds_st16(pb + PE_DLIST + ((c_dep + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm2, 0u))) << 1), k0 + lane);
          if (bigc) lds_st16(pb + PE_BLIST + ((c_big + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm2, 0u))) << 1), k0 + lane);
          if (PE_DICT && plainw) lds_st16(pb + PE_WLIST + ((c_word + __builtin_amdgcn_mbcnt_hi((uint32_t)(wm2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm2, 0u))) << 1), k0 + lane);
          lds_st32(ra, r0 | (dep << 31) | (plainw ? 1u << 28 : 0u)); lds_st32(ra + 4u, r1); lds_st32(ra + 8u, copy_x); lds_st32(ra + 12u, plainw ? wdesc : (uint32_t)dist);
          lds_st32(pb + PE_OFF + ((k0 + lane) << 2), (uint32_t)rel);
        }
        // the batch the engine's part ends in leaves the stream's state: sums up to there, the ring behind its executed pushes
        const bool last = kp_total <= k0 + K;
        if (last) {
          const uint32_t kp = my_exec;
          if (!PE_DICT && !PIPE2 && kp < 64u && ((drefmask >> kp) & 1ull) != 0ull) pe_ctl_st(pb, PEC_DSEEN, 1u);   // (the command the engine's part ends in front of)
          // (PE_DICT: the pass's last command is one whose copy is a dictionary word -- wave 0's, behind the execute: its copy length
          // is not output of this pass, its distance not one for the ring, decode.rs:2643-2644)
          const bool dlast = PE_DICT && ((dictmask >> (kp - 1u)) & 1ull) != 0ull;
          const uint32_t lit_tot = c_lit + rdlane(lit_incl, kp - 1u), t1 = rdlane(s1, kp - 1u), out_tot = c_out + rdlane(s2, kp - 1u) + rdlane(dcum, kp - 1u) - (dlast ? rdlane(copy, kp - 1u) : 0u);
          const uint32_t cmd_tot = c_cmd + (t1 & 0xFFFFu), dst_tot = c_dst + (t1 >> 16);
          const uint32_t got = (uint32_t)__popcll(pmk & ((kp >= 64u) ? ~0ull : ((1ull << kp) - 1ull)) & ~(dlast ? 1ull << (kp - 1u) : 0ull));
          if (dlast) { pe_ctl_st(pb, PEC_DICTK, k0 + kp - 1u); pe_ctl_st(pb, PEC_DICTD, rdlane((uint32_t)dist, kp - 1u)); pe_ctl_st(pb, PEC_DICTN, rdlane(copy, kp - 1u)); }
          int32_t e0 = d0, e1 = d1, e2 = d2, e3 = d3;
          if (got != 0u) {
            const uint32_t dperm = bperm(perm << 2, (uint32_t)dist);  // lane r: the distance of the r-th push
            e0 = (int32_t)rdlane(dperm, got - 1u);
            e1 = got >= 2u ? (int32_t)rdlane(dperm, got - 2u) : d0;
            e2 = got >= 3u ? (int32_t)rdlane(dperm, got - 3u) : got == 2u ? d0 : d1;
            e3 = got >= 4u ? (int32_t)rdlane(dperm, got - 4u) : got == 3u ? d0 : got == 2u ? d1 : d2;
          }
          PeStream sn = st;
          sn.P += out_tot; sn.bl0 -= lit_tot; sn.bl1 -= cmd_tot; sn.bl2 -= dst_tot; sn.quota -= out_tot; sn.mlen -= (int32_t)out_tot; sn.ncmd += cmd_tot;
          sn.d0 = e0; sn.d1 = e1; sn.d2 = e2; sn.d3 = e3;
          sn.s_cmds = cmd_tot; sn.s_lits = lit_tot; sn.s_dsts = dst_tot;   // (s_bits: wave 0, below, once it knows where the stream goes on)
          pe_st_store(pbs, sn);
          pe_ctl_st(pb, PEC_ANYDEP, c_dep + (uint32_t)__popcll(dm2)); pe_ctl_st(pb, PEC_NBIG, c_big + (uint32_t)__popcll(bm2)); pe_ctl_st(pb, PEC_NWORD, c_word + (uint32_t)__popcll(wm2));
          pe_ctl_st(pb, PEC_OUTTOT, out_tot); pe_ctl_st(pb, PEC_STAGED, (PE_STG_CAP != 0u && out_tot <= PE_STG_CAP) ? 1u : 0u);
        }
      }
      if (kp_total <= ks && T == 0u) { lds_st32(pb + PE_CTL + 4u * PEC_ANYDEP, 0u); lds_st32(pb + PE_CTL + 4u * PEC_NBIG, 0u); lds_st32(pb + PE_CTL + 4u * PEC_STAGED, 0u); lds_st32(pb + PE_CTL + 4u * PEC_OUTTOT, 0u); lds_st32(pb + PE_CTL + 4u * PEC_NWORD, 0u); }  // (nothing executed: the state stays)
      // (every wave's stores of the region before are in memory before anyone reads them as copy sources: here, a whole region's
      // tables later, the wait is over before it starts -- at the region's start it cost the stores' round trip)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PE_BAR();
      if (me == 0) {
        // where the stream goes on: the first bit of command kp_total (its head: behind the distance code, if there is one,
        // of the state it starts from)
        PeStream sn = pe_st_load(pbs);
        const uint32_t dsc = rfl(lds_ld16(pb + PE_LIST + (kp_total << 1)));
        uint32_t pbit = dsc & 0x7FFFu;
        if ((dsc >> 15) == 0u) {
          uint32_t lo, hi;
          pe_bits64(pb, pbit, lo, hi);
          const ScDist d = sc_dist(lo, hi, c.dtree, c.postfix_bits, c.num_direct);
          pbit += rfl(d.bits);
        }
        { const uint32_t nb_ = (pe_ctl_ld(pb, PEC_LBDW) << 5) + pbit; sn.s_bits = kp_total > ks ? nb_ - sn.b : 0u; if (kp_total <= ks) sn.s_cmds = 0u; sn.b = nb_; }
        // the region went through whole and the next one starts at a command with a long literal run: the next regions are the run's
        const bool dict_ends = PE_DICT && pe_ctl_ld(pb, PEC_DICTK) != 0xFFFFFFFFu;   // (the pass ends with a dictionary word still to come: no run region from here -- the next region finds the run itself)
        if (!PIPE && !dict_ends && kp_total == m && m != 0u && pbit + 64u <= c.L) PE_TRY_RUN(sn, pbit);
        // ... or the region listed nothing because its first command's literal run is more than its path holds (4300 literals of
        // 7.5 bits) though less than PE_RUN_MIN: regions of its own all the same -- giving the command back would keep the engine
        // away from the commands behind it too (seen on the high-entropy streams: the rest of a metablock on one wave)
        if (!PIPE && m == 0u && pbit + 64u <= c.L) PE_TRY_RUN_FROM(sn, pbit, 1024u);
        pe_ctl_st(pb, PEC_NEXT_LBDW, sn.b >> 5);
        // an invocation goes on with the next region while whole regions go through; anything else is the checked loop's
        bool cont = (kp_total == m && m != 0u) || (!PIPE && m == 0u && sn.run_on != 0u);
        if (!PIPE && pe_ctl_ld(pb, PEC_OVF) >= 3u && sn.run_on == 0u) cont = false;   // (closure-bound: see the records)
        if (PIPE && cont && pbit + 64u <= c.L) {
          // (two engines: a command whose literal run wants regions of its own ends the invocation in front of it -- the one-engine
          // form has those regions, and the caller is told to take it next)
          uint32_t lo_, hi_;
          pe_bits64(pb, pbit, lo_, hi_);
          const ScHead h_ = sc_head(lo_, hi_, c.cmd_tree, c.lut_vgpr);
          if (rfl(h_.insert) >= (REMOTE ? PE_RUN_MIN : PE_PIPE_DECLINE)) { cont = false; pe_ctl_st(pbs, PEC_DECLINE, 1u); }   // (a gang's regions are whole ones: what one of them holds, it takes)
        }
        if (REMOTE && wn + 64u > PE_WCAP) {
          // (a gang: the closure has filled its room -- a stretch of few literals; the one-block form halves its regions there and hands such
          // streams to the scan engine, this form has no room left for the entry's states: the rest of the metablock is the one-block form's)
          cont = false; pe_ctl_st(pbs, PEC_DECLINE, 3u);
        }
        if (REMOTE && m == 0u && pbit + 64u <= c.L) {
          // (a gang: the region listed nothing -- as a rule its first command's literal run is more than what is left of the window holds: the
          // one-block form's, which gives such a run regions of its own; an invocation that takes nothing and does not say why sends the stream
          // through the one-wave loop for a while -- seen on a 64 MiB stream's seed: a tenth of the commands, two thirds of the time)
          uint32_t lo_, hi_;
          pe_bits64(pb, pbit, lo_, hi_);
          const ScHead h_ = sc_head(lo_, hi_, c.cmd_tree, c.lut_vgpr);
          if (rfl(h_.insert) >= 1024u) pe_ctl_st(pbs, PEC_DECLINE, pe_ctl_ld(pbs, PEC_DECLINE) | 1u);
        }
        if (REMOTE && !cont) GANG_STAT(gc, 37, 1);
        if (REMOTE && !cont && kp_total != m) GANG_STAT(gc, 35, 1);
#ifdef BROTLI_AMD_PE_DEBUG
        if (PIPE && blockIdx.x == 0 && lane == 0) printf("   region %u: %u commands listed, %u executed, goes on at %u, cont %u, P now %llu ncmd %u\n", kseq, m, kp_total, sn.b, cont ? 1u : 0u, (unsigned long long)sn.P, sn.ncmd);
#endif
        if (REMOTE && kseq + 1u >= GC_MAX_REGIONS) cont = false;   // (the tags of the state's granules count regions in twelve bits)
        pe_ctl_st(pb, PEC_CONT, cont ? 1u : 0u);  // (a word of its own: wave 0 writes PEC_GO for the next region while the others may still be here)
        pe_st_store(pbs, sn);
        if (!PIPE) { lds_sync(); pe_ctl_st(pb, PEC_NXOK, rseq); }
        if (REMOTE) {
          // the stream's state is the next region's from here on: granule by granule, each with the tag its reader waits for; the
          // invocation's end in a word of its own behind them (whoever waits for a region that will not come looks at it)
          pe_ctl_st(pb, PEC_MYNEXT, sn.b);
          lds_sync();
          const uint32_t v = lane < 25u ? *reinterpret_cast<lds_vu32*>(&g_smem[pbs + PE_CTL + 4u * (PEC_STATE + (lane < 25u ? lane : 0u))]) : lane == 25u ? (cont ? 1u : 0u) | ((pe_ctl_ld(pbs, PEC_DECLINE) & 3u) << 1) | (pe_ctl_ld(pb, PEC_DSEEN) != 0u ? 8u : 0u) : lane == 27u ? pe_ctl_ld(pb, PEC_PREVOUT) : pe_ctl_ld(pb, PEC_OUTTOT);
          if (lane < GC_STATE_WORDS) gang_st64(gc, GC_STATE + 8u * lane, (uint64_t)v | ((uint64_t)((epoch << 12) | (kseq + 1u)) << 32));
          if (!cont) { gang_drain(); if (lane == 0u) gang_st64(gc, GC_STOP, ((uint64_t)epoch << 32) | (uint64_t)(kseq + 1u)); }
        }
        if (PIPE2) {
          pe_ctl_st(pb, PEC_MYNEXT, sn.b);
          // the stream's state is the next region's from here on; the invocation's end is everybody's to know first
          lds_sync();
          if (!cont) pe_ctl_st(pbs, PEC_STOP, 1u);
          lds_sync();
          pe_ctl_st(pbs, PEC_RESOLVED, kseq + 1u);
        }
      }
    }
    PE_PROF(11);
    if (REMOTE && me == 0) GANG_STAT(gc, 26, __builtin_amdgcn_s_memtime() - gs_arr);   // .. resolve done (wave 0 past the publish)
    if (REMOTE && me == 0) GT(4);
    // (a gang) Two waits for other CUs' output: here for the regions up to the one before the region before -- the literals, and the copies that
    // read nothing younger, start at once --, and behind them for the region before, whose last bytes only the copies marked `dep` in the resolve
    // read (with uniform distances a handful a region: they go with the copies that read this region's own output).  The word is another CU's:
    // wave 0 looks at it, lets this CU forget what it has cached of the output, and tells the others.
    // Which of the two it is, every engine decides for itself from what it has seen: two waits cost a second fence and put more copies on the
    // slower road, and pay where the executes are what the stream waits for (many commands a region: +8 % on one stream of 64 MiB and more,
    // -4 .. -10 % on batches of the metric's 4 MiB streams if it were the rule) -- an engine that has waited long for the region before's
    // output waits twice from the next region on, one whose two waits were short goes back to one.  (Gangs of eight only: smaller ones wait for a
    // block that is busy, not for a chain -- measured, two waits cost them 5 .. 8 %.)
    auto await_output = [&](const uint32_t upto, const uint32_t stat) -> uint32_t {
      uint32_t spins = 0; (void)spins; (void)stat;
      const uint64_t t0_ = __builtin_amdgcn_s_memtime();
      for (;;) {
        const uint64_t ew = gang_ld64(gc, GC_EXEC);
        if ((uint32_t)(ew >> 32) == epoch && (uint32_t)ew >= upto) break;
        __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins);
      }
      const uint32_t waited = (uint32_t)(__builtin_amdgcn_s_memtime() - t0_);
      GANG_STAT(gc, stat, waited);
      gang_acquire();
      return waited;
    };
    const bool relaxed = REMOTE && pe_ctl_ld(pb, PEC_LAG) != 0u;   // (the resolve marked the copies that read the region before's output)
    uint32_t waited1 = 0; (void)waited1;
    if (REMOTE && me == 0) {
      if (relaxed) {
        // (the resolve marked the copies that read the one or the two regions before: as many executes besides this one may be under way)
        const uint32_t depth = pe_ctl_ld(pb, PEC_DEPTH);
        if (kseq >= depth + 1u) waited1 = await_output(kseq - depth, 9u);
        if (depth == 1u && waited1 > 6000u && gang_m >= PE_DEPTH2_MIN) pe_ctl_st(pb, PEC_RELAX, 2u);
      }
      else if (kseq != 0u) { waited1 = await_output(kseq, 8u); if (waited1 > 6000u && gang_m >= 8u) pe_ctl_st(pb, PEC_RELAX, 1u); GANG_STAT(gc, 30, 1); }
      lds_sync();
      pe_ctl_st(pbs, PEC_EXECUTED, kseq);
      GT(5);
    }
    if (PIPE) {
      // the region before's output is in memory before this one's copies read it (its engine says so)
      uint32_t spins = 0; (void)spins;
      while (pe_ctl_ld(pbs, PEC_EXECUTED) < kseq) { __builtin_amdgcn_s_sleep(2); PE_SPIN_CHECK(spins); }
      PE_PROF(14);
    }
    // ---- execute ----
    RG_STAMP(1);   // resolve done
    {
      gu8* const o = out + P0;
      // (a) lane = command: the literals in front of the path, decoded again one after the other; the literals on the path out
      // of lit[], four bytes a step; the command's
      // copy where it is short and its source lies in front of the region's output (one 16-byte load, stores in pieces).
      // Every wave the batch it resolved -- and, where the region has at most eight batches, the path's literals of batch b by
      // wave b + 9 (whose own batch does not exist) at the same time: the records and offsets are in LDS behind the barrier above.
      // Where the region's whole output fits J1's room (nobody reads J1 or NEXT8 any more), it is put together THERE and written
      // out in one piece at the end: the stores of (a) are a few bytes each at addresses all over the place, and the copies
      // of (c) wait for each other -- through LDS both cost a fraction of what they cost through memory.
      const bool staged = pe_ctl_ld(pb, PEC_STAGED) != 0u;
      const uint32_t sg = pb + PE_STG + ((uint32_t)P0 & 15u);   // (the stage as the output lies in memory, modulo sixteen: see write_out; J1's room has 64 bytes to spare)
      // (a wave's LZ77 copy into the stage, decode.rs:2641-2720: byte q comes from byte q mod distance of the `distance` bytes
      // in front of the destination, which lie in the stage or, in front of the region, in memory)
      auto stage_copy = [&](const uint32_t dpos, const uint32_t n, const uint32_t dist) {
        uint32_t mm = dist >= 64u ? lane : lane % dist; const uint32_t step = dist > 64u ? 64u : 64u % dist;
        for (uint32_t q = 0; q < n; q += 64u) {
          if (q + lane < n) {
            const int32_t sp = (int32_t)dpos - (int32_t)dist + (int32_t)(dist >= n ? q + lane : mm);
            const uint32_t t = sp < 0 ? (uint32_t)*(o + (int64_t)sp) : lds_ld8(sg + (uint32_t)sp);
            lds_st8(sg + dpos + q + lane, t);
          }
          mm += step; if (mm >= dist) mm -= dist;
        }
      };
      const uint32_t kp_all = pe_ctl_ld(pb, PEC_KP);
#ifdef BROTLI_AMD_PE_NO_EXEC_SPLIT
      const bool exec_split = false;
#else
      const bool exec_split = nb <= GW / 2u;
#endif
      auto path_literals = [&](const uint32_t b, const uint32_t cnt) {
        const uint32_t k = (b << 6) + lane;
        const bool on = lane < cnt && k >= ks;
        const uint32_t ra = pb + PE_REC + ((on ? k : 0u) << 4);
        const uint32_t r0 = lds_ld32(ra), r1 = lds_ld32(ra + 4u);
        const uint32_t off = lds_ld32(pb + PE_OFF + ((on ? k : 0u) << 2));
        const uint32_t ins = on ? r1 & 0xFFFFu : 0u, ry = r1 >> 16;
        uint32_t u = (r0 >> 15) & 255u;
        u = u < ins ? u : ins;
        gu8* dst = o + off + u;
        uint32_t n = ins - u; n = n <= PE_LANE_LITS ? n : 0u;   // (longer runs: a wave of their own, below)
        uint32_t la = pb + PE_LIT + ry;
        if (staged) {
          uint32_t dx = sg + off + u;
          while (__ballot(n != 0u) != 0ull) {
            const uint32_t b0 = lds_ld8(la), b1 = lds_ld8(la + 1u), b2 = lds_ld8(la + 2u), b3 = lds_ld8(la + 3u);
            if (n != 0u) { lds_st8(dx, b0); if (n > 1u) lds_st8(dx + 1u, b1); if (n > 2u) lds_st8(dx + 2u, b2); if (n > 3u) lds_st8(dx + 3u, b3); }
            const uint32_t adv = n < 4u ? n : 4u;
            dx += adv; la += adv; n -= adv;
          }
          return;
        }
        while (__ballot(n != 0u) != 0ull) {
          const uint32_t b0 = lds_ld8(la), b1 = lds_ld8(la + 1u), b2 = lds_ld8(la + 2u), b3 = lds_ld8(la + 3u);
          if (n >= 4u) { *reinterpret_cast<gu32*>(dst) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24); dst += 4; la += 4u; n -= 4u; }
          else if (n != 0u) { dst[0] = (uint8_t)b0; if (n > 1u) dst[1] = (uint8_t)b1; if (n > 2u) dst[2] = (uint8_t)b2; n = 0u; }
        }
      };
      if (my_exec != 0u) {
        const uint32_t k = (bw << 6) + lane;
        const bool on = lane < my_exec && k >= ks;
        const uint32_t ra = pb + PE_REC + ((on ? k : 0u) << 4);
        const uint32_t r0 = lds_ld32(ra), r1 = lds_ld32(ra + 4u), cn = lds_ld32(ra + 8u), dist = lds_ld32(ra + 12u);
        const uint32_t off = lds_ld32(pb + PE_OFF + ((on ? k : 0u) << 2));
        const uint32_t ins = on ? r1 & 0xFFFFu : 0u;
        uint32_t u = (r0 >> 15) & 255u;
        u = u < ins ? u : ins;
        // (the region's quota check leaves SC_MIN_QUOTA bytes of room behind P0: sixteen bytes from a source in front of it are inside the buffer)
        const bool shortcopy = on && (r0 >> 31) == 0u && cn != 0u && cn <= PE_LANE_COPY;
        gu8* const cdst = o + off + ins;
        u32x4 cv[PE_LANE_COPY / 16u];
        _Pragma("unroll") for (uint32_t g = 0; g < PE_LANE_COPY / 16u; g++) {
          cv[g] = u32x4{0u, 0u, 0u, 0u};
          if (shortcopy && cn > 16u * g) cv[g] = *reinterpret_cast<gu32x4*>(cdst - dist + 16u * g);
        }
        uint32_t y = r0 & 0x7FFFu;
        gu8* dst = o + off;
        uint32_t dx = sg + off;
        while (__ballot(u != 0u) != 0ull) {
          uint32_t sy, ln;
          sc_lookup(c.lit_tree, pe_bits32(pb, u != 0u ? y : 0u), sy, ln);
          if (u != 0u) { if (staged) lds_st8(dx, sy); else *dst = (uint8_t)sy; dst++; dx++; y += ln; u--; }
        }
        if (staged) {
          // the short copy's bytes into the stage, one at a time out of the sixteen loaded
          uint32_t w0 = cv[0].x, w1 = cv[0].y, w2 = cv[0].z, w3 = cv[0].w, left = shortcopy ? cn : 0u, cx = sg + off + ins;
          while (__ballot(left != 0u) != 0ull) {
            if (left != 0u) { lds_st8(cx, w0 & 0xFFu); cx++; left--; }
            w0 = __builtin_amdgcn_alignbit(w1, w0, 8); w1 = __builtin_amdgcn_alignbit(w2, w1, 8); w2 = __builtin_amdgcn_alignbit(w3, w2, 8); w3 >>= 8;
          }
        } else if (shortcopy) {
          _Pragma("unroll") for (uint32_t g = 0; g < PE_LANE_COPY / 16u; g++) {
            const uint32_t w[4] = {cv[g].x, cv[g].y, cv[g].z, cv[g].w};
            _Pragma("unroll") for (uint32_t q4 = 0; q4 < 4u; q4++) {
              const uint32_t q = 4u * g + q4;
              if (cn >= 4u * q + 4u) *reinterpret_cast<gu32*>(cdst + 4u * q) = w[q4];
              else if (cn > 4u * q) {
                const uint32_t rest = cn - 4u * q;
                cdst[4u * q] = (uint8_t)w[q4];
                if (rest > 1u) cdst[4u * q + 1u] = (uint8_t)(w[q4] >> 8);
                if (rest > 2u) cdst[4u * q + 2u] = (uint8_t)(w[q4] >> 16);
              }
            }
          }
        }
        if (!exec_split || bw >= GW / 2u - 1u) path_literals(bw, my_exec);
      } else if (exec_split && bw >= GW / 2u && bw < GW - 1u) {
        const uint32_t b = bw - GW / 2u;
        const uint32_t cnt = kp_all <= (b << 6) ? 0u : (kp_all - (b << 6) < 64u ? kp_all - (b << 6) : 64u);
        if (cnt != 0u) path_literals(b, cnt);
      }
      // (no barrier: what (b) stores lies elsewhere, and a wave that is through with (a) -- most have no batch -- takes items at once;
      // (c) waits for both)
      PE_PROF(8);
      RG_STAMP(2);   // (a) done
      PE_COUNT(28, kp);
      if (!PIPE) {   // (wave 0 says where the stream goes on behind the resolve's last barrier, while the others execute: as a rule long since)
        while (pe_ctl_ld(pb, PEC_NXOK) != rseq) __builtin_amdgcn_s_sleep(1);
        lds_sync();
      }
      if (!PIPE && pe_ctl_ld(pb, PEC_CONT) != 0u) {  // the next region's input is on its way while the rest of this one is executed
        const uint32_t nl = pe_ctl_ld(pb, PEC_NEXT_LBDW);
        pre_a = nl + T < limit_dw ? in_dw[nl + T] : 0u; pre_b = (T < 6u && nl + PE_CHUNKS + T < limit_dw) ? in_dw[nl + PE_CHUNKS + T] : 0u;
        pre_ok = true;
      }
      // (b) the commands listed for it get a wave each: long literal runs out of lit[], long copies whose source lies in front
      // of the region's output.  A wave takes its items three at a time: the copies of up to 64 bytes are asked for side by
      // side and stored when all three are there (their sources lie far back: a round trip to memory each).
      {
        const uint32_t nbig = pe_ctl_ld(pb, PEC_NBIG);
        PE_COUNT(12, nbig);
        auto big_item = [&](const uint32_t j, uint32_t& pt, uint32_t& pd, uint32_t& pn) {
          pn = 0; pt = 0; pd = 0;
          if (j >= nbig) return;
          const uint32_t k = rfl(lds_ld16(pb + PE_BLIST + (j << 1)));
          const uint32_t ra = pb + PE_REC + (k << 4);
          const uint32_t r0 = rfl(lds_ld32(ra)), r1 = rfl(lds_ld32(ra + 4u)), cn = rfl(lds_ld32(ra + 8u)), dist = rfl(lds_ld32(ra + 12u));
          const uint32_t off = rfl(lds_ld32(pb + PE_OFF + (k << 2)));
          const uint32_t ins = r1 & 0xFFFFu, ry = r1 >> 16;
          uint32_t u = (r0 >> 15) & 255u; u = u < ins ? u : ins;
          const uint32_t n = ins - u;
          if (n > PE_LANE_LITS) {
            gu8* const lp = o + off + u;
            const uint32_t la = pb + PE_LIT + ry;
            if (staged) { for (uint32_t i = lane; i < n; i += 64u) lds_st8(sg + off + u + i, lds_ld8(la + i)); }
            else for (uint32_t i = lane; i < n; i += 64u) lp[i] = (uint8_t)lds_ld8(la + i);
          }
          if (cn > PE_LANE_COPY && (r0 >> 31) == 0u) {
            gu8* const dst = o + off + ins; gu8* const src = dst - dist;   // (all of the source in front of the region)
            if (cn <= 64u) { if (lane < cn) pt = src[lane]; pd = off + ins; pn = cn; }
            else if (staged) { for (uint32_t q = lane; q < cn; q += 64u) lds_st8(sg + off + ins + q, src[q]); }
            else {
              const uint32_t n16 = cn >> 4;
              for (uint32_t q = lane; q < n16; q += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)q * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)q * 16);
              const uint32_t tail = n16 << 4;
              if (tail + lane < cn) dst[tail + lane] = src[tail + lane];
            }
          }
        };
        auto big_store = [&](const uint32_t pt, const uint32_t pd, const uint32_t pn) {
          if (lane < pn) { if (staged) lds_st8(sg + pd + lane, pt); else o[pd + lane] = (uint8_t)pt; }
        };
        for (;;) {   // (a shared counter hands the items out, three at a time)
          const uint32_t j0 = pe_atomic_add_uniform(pb + PE_CTL + 4u * PEC_BIGNEXT, 3u);
          if (j0 >= nbig) break;
          uint32_t t0, d0, n0, t1, d1, n1, t2, d2, n2;
          big_item(j0, t0, d0, n0); big_item(j0 + 1u, t1, d1, n1); big_item(j0 + 2u, t2, d2, n2);
          big_store(t0, d0, n0); big_store(t1, d1, n1); big_store(t2, d2, n2);
        }
      }
#if PE_DICT
      {
        // (b') the pass's words of the static dictionary (decode.rs:2593-2640, transform.rs:737-795), a wave each: lane = byte of the word
        const uint32_t nword = pe_ctl_ld(pb, PEC_NWORD);
        if (nword != 0u) for (;;) {
          const uint32_t j = pe_atomic_add_uniform(pb + PE_CTL + 4u * PEC_WNEXT, 1u);
          if (j >= nword) break;
          const uint32_t k = rfl(lds_ld16(pb + PE_WLIST + (j << 1)));
          const uint32_t ra = pb + PE_REC + (k << 4);
          const uint32_t r1 = rfl(lds_ld32(ra + 4u)), wd = rfl(lds_ld32(ra + 12u)), at = rfl(lds_ld32(pb + PE_OFF + (k << 2))) + (r1 & 0xFFFFu);
          const WordShape w = word_shape(wd >> 24, (wd >> 17) & 127u);
          const uint32_t ob = dictionary_word_bytes(dict, wd & 0x1FFFFu, w, pb + PE_CB + me * 128u);   // (the chunks' rank bases are nobody's any more: 128 bytes a wave for the transforms that walk the word)
          if (lane < w.total) { if (staged) lds_st8(sg + at + lane, ob); else o[at + lane] = (uint8_t)ob; }
        }
      }
#endif
      PE_PROF(9);
      RG_STAMP(3);   // (b) done (this wave's share)
      if (REMOTE && me == 0) GT(8);
      // (a gang whose executes wait twice) The second wait -- for the region before's output -- stands in front of the word that says this region's is
      // there, and in front of the copies that read that output (`lagging` ones: their source begins in front of the region) and of those that build on them.
      auto second_wait = [&]() {
        if (REMOTE && me == 0 && kseq != 0u && relaxed) {
          const uint32_t waited2 = await_output(kseq, 8u);
          if (waited1 + waited2 < 1500u) pe_ctl_st(pb, PEC_RELAX, pe_ctl_ld(pb, PEC_DEPTH) - 1u);   // (both short: one execute fewer in flight will do)
        }
        if (REMOTE && me == 0) GT(9);
      };
      // (c) copies that read the region's own output (`dependent` ones; a gang whose executes wait twice: and the `lagging` ones, whose source begins in
      // front of the region -- in the region before's output, which may still be on its way).  Through round 5 those that build on another one went one
      // after the other on one wave, a memory round trip each (a gang's region of the copy part: 19 of 241, 28 K clocks behind the second wait).  Now by
      // LEVELS: a copy's level is one more than the highest level among the earlier dependent copies whose destination its source touches (destinations
      // lie in command order: two binary searches give that range [a, b) of the list), 0 where there is none; a level's copies are independent of each
      // other and go side by side, loads first, then stores.  The levels settle by relaxation over the list (they only grow; after r rounds every copy
      // of level < r has its own): PE_DEP_ROUNDS rounds at most, what is deeper goes in order behind the rest, as before.  Lagging spreads the same way
      // (bit 7): what neither lags nor builds on a copy that does is done in front of the second wait.
      const uint32_t ndep = pe_ctl_ld(pb, PEC_ANYDEP);
      PE_COUNT(13, ndep);
      auto dependent_copies = [&]() {
        const bool lagging = REMOTE && relaxed && kseq != 0u;
        const uint32_t DA = pb + PE_POR, DB = DA + 2u * PE_CMDS, DL = DB + 2u * PE_CMDS;   // (the ranks' room: nobody reads a rank behind the details)
        for (uint32_t j = T; j < ndep; j += 64u * GW) {
          const uint32_t k = lds_ld16(pb + PE_DLIST + (j << 1));
          const uint32_t ra = pb + PE_REC + (k << 4);
          const uint32_t cn = lds_ld32(ra + 8u), dist = lds_ld32(ra + 12u);
          const uint32_t dst = lds_ld32(pb + PE_OFF + (k << 2)) + (lds_ld32(ra + 4u) & 0xFFFFu);
          // the source's part inside the region (a copy that repeats itself: what lies in front of its destination)
          const uint32_t s_lo = dist <= dst ? dst - dist : 0u, s_hi = dist >= cn ? (dst + cn > dist ? dst + cn - dist : 0u) : dst;
          uint32_t a = j, b = j;
          if (s_hi > s_lo) {
            uint32_t lo_ = 0, hi_ = j;   // the first earlier dependent copy whose destination ends behind s_lo
            while (lo_ < hi_) {
              const uint32_t mid = (lo_ + hi_) >> 1;
              const uint32_t km = lds_ld16(pb + PE_DLIST + (mid << 1));
              const uint32_t de = lds_ld32(pb + PE_OFF + (km << 2)) + (lds_ld32(pb + PE_REC + (km << 4) + 4u) & 0xFFFFu) + lds_ld32(pb + PE_REC + (km << 4) + 8u);
              if (de > s_lo) hi_ = mid; else lo_ = mid + 1u;
            }
            a = lo_; hi_ = j;            // ... and the first from there on whose destination begins at s_hi or behind it
            while (lo_ < hi_) {
              const uint32_t mid = (lo_ + hi_) >> 1;
              const uint32_t km = lds_ld16(pb + PE_DLIST + (mid << 1));
              const uint32_t ds = lds_ld32(pb + PE_OFF + (km << 2)) + (lds_ld32(pb + PE_REC + (km << 4) + 4u) & 0xFFFFu);
              if (ds >= s_hi) hi_ = mid; else lo_ = mid + 1u;
            }
            b = lo_;
          }
          lds_st16(DA + (j << 1), a); lds_st16(DB + (j << 1), b);
          lds_st8(DL + j, (a < b ? 1u : 0u) | ((lagging && dist > dst) ? 0x80u : 0u));
        }
        // this wave's copies: lane l has entry me + GW l of the list in its registers
        const uint32_t myj = me + GW * lane;
        const bool have = myj < ndep;
        const uint32_t e_k = have ? lds_ld16(pb + PE_DLIST + (myj << 1)) : 0u;
        const uint32_t e_ra = pb + PE_REC + (e_k << 4);
        const uint32_t e_n = lds_ld32(e_ra + 8u), e_d = lds_ld32(e_ra + 12u), e_p = lds_ld32(pb + PE_OFF + (e_k << 2)) + (lds_ld32(e_ra + 4u) & 0xFFFFu);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PE_BAR();
        RG_STAMP(4);   // the dependent copies' ranges, everybody's (a) and (b) in memory
        if (REMOTE && me == 0) GT(10);
        for (uint32_t round = 0; round < PE_DEP_ROUNDS; round++) {   // (bit r of PEC_DEPCHG: round r changed a level -- written in round r only, read behind its barrier)
          bool chg = false;
          for (uint32_t j = T; j < ndep; j += 64u * GW) {
            const uint32_t a = lds_ld16(DA + (j << 1)), b = lds_ld16(DB + (j << 1)), cur = lds_ld8(DL + j);
            uint32_t mx = 0, lg = cur & 0x80u;
            for (uint32_t i2 = a; i2 < b; i2++) { const uint32_t v = lds_ld8(DL + i2); mx = (v & 0x7Fu) > mx ? (v & 0x7Fu) : mx; lg |= v & 0x80u; }
            const uint32_t nv = a < b ? ((mx + 1u < 0x7Fu ? mx + 1u : 0x7Fu) | lg) : cur;
            if (nv != cur) { lds_st8(DL + j, nv); chg = true; }
          }
          if (__ballot(chg) != 0ull && lane == 0) pe_atomic_or(pb + PE_CTL + 4u * PEC_DEPCHG, 1u << round);
          PE_BAR();
          if (((pe_ctl_ld(pb, PEC_DEPCHG) >> round) & 1u) == 0u) break;
        }
        // what this wave's copies are: level | lagging << 7; those of level PE_DEP_ROUNDS and more are the last wave's, in order (bit 29 of w0: not)
        const uint32_t e_l = have ? lds_ld8(DL + myj) : 0xFFu;
        const bool shallow = have && (e_l & 0x7Fu) < PE_DEP_ROUNDS;
        if (shallow) lds_st32(e_ra, lds_ld32(e_ra) | (1u << 29));
        { const uint64_t sh0 = __ballot(shallow && (e_l & 0x80u) == 0u), sh1 = __ballot(shallow && (e_l & 0x80u) != 0u), deep = __ballot(have && !shallow);
          uint32_t m0 = 0, m1 = 0;
          for (uint64_t q = sh0; q; q &= q - 1ull) { const uint32_t v = rdlane(e_l, (uint32_t)__builtin_ctzll(q)) & 0x7Fu; m0 = v + 1u > m0 ? v + 1u : m0; }
          for (uint64_t q = sh1; q; q &= q - 1ull) { const uint32_t v = rdlane(e_l, (uint32_t)__builtin_ctzll(q)) & 0x7Fu; m1 = v + 1u > m1 ? v + 1u : m1; }
          if (lane == 0) { if (m0) pe_atomic_max(pb + PE_CTL + 4u * PEC_DEPLV0, m0); if (m1) pe_atomic_max(pb + PE_CTL + 4u * PEC_DEPLV1, m1); if (deep) pe_atomic_max(pb + PE_CTL + 4u * PEC_DEPDEEP, 1u); } }
        PE_BAR();
        const uint32_t nlv0 = pe_ctl_ld(pb, PEC_DEPLV0), nlv1 = pe_ctl_ld(pb, PEC_DEPLV1);   // levels (their number) without and with lagging
#ifdef BROTLI_AMD_GANG_TRACE
        { for (uint32_t q = 0; q < 4u; q++) { const uint32_t c_ = (uint32_t)__popcll(__ballot(shallow && e_l == (0x80u | q))); if (c_ && lane == 0) pe_atomic_add_uniform(pb + PE_CTL + 4u * (120u + q), c_); }
          if (T == 0u) pe_ctl_st(pb, 124u, nlv0 | (nlv1 << 8) | ((pe_ctl_ld(pb, PEC_DEPDEEP) != 0u ? 1u : 0u) << 16)); }
#endif
        const bool any_deep = pe_ctl_ld(pb, PEC_DEPDEEP) != 0u;
        auto one_copy = [&](const uint32_t dpos, const uint32_t n, const uint32_t dist) {
          if (staged) { stage_copy(dpos, n, dist); return; }
          gu8* const dst = o + dpos; gu8* const src = dst - dist;
          if (dist < n) {
            // the copy overlaps itself (decode.rs:2657-2663, 2690-2720: byte by byte, so a pattern of `dist` bytes repeats)
            if (dist >= 64u) { for (uint32_t q = lane; q < n + lane; q += 64u) if (q < n) dst[q] = src[q]; }  // a step reads what earlier steps wrote
            else {
              uint32_t mm = lane % dist; const uint32_t step = 64u % dist;
              for (uint32_t q = 0; q < n; q += 64u) { if (q + lane < n) dst[q + lane] = src[mm]; mm += step; if (mm >= dist) mm -= dist; }
            }
          } else if (n <= 64u) { uint32_t t = 0; if (lane < n) t = src[lane]; if (lane < n) dst[lane] = (uint8_t)t; }
          else {
            const uint32_t n16 = n >> 4;
            for (uint32_t q = lane; q < n16; q += 64u) *reinterpret_cast<gu32x4*>(dst + (uint64_t)q * 16) = *reinterpret_cast<gu32x4*>(src + (uint64_t)q * 16);
            const uint32_t tail = n16 << 4;
            if (tail + lane < n) dst[tail + lane] = src[tail + lane];
          }
        };
        auto level = [&](const uint32_t code) {   // this wave's copies of one level: four loads in flight where they are plain ones of at most 64 bytes
          uint64_t mm = __ballot(shallow && e_l == code);
          while (mm) {
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, n0 = 0, n1 = 0, n2 = 0, n3 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            auto take = [&](uint32_t& t, uint32_t& pn, uint32_t& pp) {
              if (!mm) return;
              const uint32_t i2 = (uint32_t)__builtin_ctzll(mm);
              mm &= mm - 1ull;
              const uint32_t n = rdlane(e_n, i2), dist = rdlane(e_d, i2), dpos = rdlane(e_p, i2);
              if (!staged && dist >= n && n <= 64u) { if (lane < n) t = *(o + dpos - dist + lane); pn = n; pp = dpos; }
              else one_copy(dpos, n, dist);
            };
            take(t0, n0, p0); take(t1, n1, p1); take(t2, n2, p2); take(t3, n3, p3);
            if (lane < n0) o[p0 + lane] = (uint8_t)t0;
            if (lane < n1) o[p1 + lane] = (uint8_t)t1;
            if (lane < n2) o[p2 + lane] = (uint8_t)t2;
            if (lane < n3) o[p3 + lane] = (uint8_t)t3;
          }
        };
        // the copies of level PE_DEP_ROUNDS and more (and, a gang: whatever they build on may lag), in command order
        auto in_order = [&]() {
          const uint32_t kp = kp_all;
          for (uint32_t k0 = ks & ~63u; k0 < kp; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const uint32_t ra = pb + PE_REC + ((k < kp ? k : 0u) << 4);
            const uint32_t x0 = lds_ld32(ra), x1 = lds_ld32(ra + 4u), xn = lds_ld32(ra + 8u), xd = lds_ld32(ra + 12u), xo = lds_ld32(pb + PE_OFF + ((k < kp ? k : 0u) << 2));
            uint64_t dm = __ballot(k >= ks && k < kp && (x0 >> 31) != 0u && ((x0 >> 29) & 1u) == 0u);
            while (dm) {
              const uint32_t kk = (uint32_t)__builtin_ctzll(dm);
              dm &= dm - 1ull;
              one_copy(rdlane(xo, kk) + (rdlane(x1, kk) & 0xFFFFu), rdlane(xn, kk), rdlane(xd, kk));
            }
          }
        };
        for (uint32_t lv = 0; lv < nlv0; lv++) {
          level(lv);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PE_BAR();
        }
        RG_STAMP(5);   // what does not wait for the region before is done
        if (REMOTE && me == 0) GT(11);
        second_wait();
        if (nlv1 != 0u || any_deep) PE_BAR();
        for (uint32_t lv = 0; lv < nlv1; lv++) {
          level(lv | 0x80u);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PE_BAR();
          if (REMOTE && me == 0 && lv == 0u) GT(12);
        }
        if (REMOTE && me == 0) GT(14);
        if (any_deep && me == GW - 1u) in_order();
      };
      if (ndep != 0u) dependent_copies(); else second_wait();
      if (staged) {
        // the region's output, out of the stage in one piece: sixteen bytes a thread and step
        PE_BAR();
        if (REMOTE && me == 0) GT(12);
        const uint32_t tot = pe_ctl_ld(pb, PEC_OUTTOT);
        write_out(sg, o, tot);
        if (REMOTE && me == 0) GT(13);
      }
      PE_PROF(10);
    }
#if PE_DICT
    if (pe_ctl_ld(pb, PEC_DICTK) != 0xFFFFFFFFu) {
      // the pass ended behind the literals of a command whose copy is a word of the static dictionary (decode.rs:2593-2640, as
      // lean_rec_commands takes them): wave 0 puts it behind them, and the commands behind it get a pass of their own
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PE_BAR();   // (the pass's output is complete, nobody reads its records any more)
      if (me == 0) { pe_dict_word(pbs, pb, out, dict, m); pe_ctl_st(pb, PEC_DCAND, 0u); }
      PE_BAR();
      if (pe_ctl_ld(pb, PEC_AGAIN) != 0u) {
        ks = pe_ctl_ld(pb, PEC_KS);
        P0 = (uint64_t)pe_ctl_ld(pb, PEC_P0_LO) | ((uint64_t)pe_ctl_ld(pb, PEC_P0_HI) << 32);
        rseq++;
        goto pe_pass;
      }
    }
#endif
    if (PIPE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PE_BAR();
      if (PIPE2 && T == 0u) { lds_sync(); pe_ctl_st(pbs, PEC_EXECUTED, kseq + 1u); }
      if (REMOTE && T == 0u) { const uint64_t t0_ = __builtin_amdgcn_s_memtime(); (void)t0_; GANG_STAT(gc, 27, t0_ - gs_arr); gang_release();
 GANG_STAT(gc, 17, __builtin_amdgcn_s_memtime() - t0_); gang_st64(gc, GC_EXEC, ((uint64_t)epoch << 32) | (uint64_t)(kseq + 1u)); }
#ifdef BROTLI_AMD_GANG_TRACE
      if (REMOTE && T == 0u && epoch == (uint32_t)(BROTLI_AMD_GANG_TRACE) && kseq < 64u) { GT(6);   // (the last 8 KiB of the arena's image: a block of sixteen waves has 40 KiB of arena)
        gt_ts[13] = (gt_ts[6] & ~0xFFFFFFFFull) | pe_ctl_ld(pb, 124u);
        pe_ctl_st(pb, 120u, 0u); pe_ctl_st(pb, 121u, 0u); pe_ctl_st(pb, 122u, 0u); pe_ctl_st(pb, 123u, 0u);
        for (uint32_t q = 0; q < 15u; q++) gang_st64(gc, GC_ARENA + (40u << 10) + 128u * kseq + 8u * q, gt_ts[q]);
        gang_st64(gc, GC_ARENA + (40u << 10) + 128u * kseq + 120u, (uint64_t)pe_ctl_ld(pb, PEC_KP) | ((uint64_t)blockIdx.x << 32) | ((uint64_t)pe_ctl_ld(pb, PEC_ANYDEP) << 16)); }
#endif
    }
  };
#if !PE_CFG_PIPE && !PE_CFG_REMOTE
  for (;;) {
    if (me == 0) {
      const PeStream st = pe_st_load(pbs);
      const uint32_t lbdw_ = st.b >> 5;
      const uint32_t avail = in_limit - (lbdw_ << 5);
      const bool go = td_ok && st.b < in_limit && avail >= PE_MIN_INPUT && st.quota >= SC_MIN_QUOTA && (st.bl1 != 0u || st.run_on != 0u);
      pe_ctl_st(pb, PEC_GO, go ? 1u : 0u);
#ifdef BROTLI_AMD_PE_DEBUG
      if (blockIdx.x == 0 && lane == 0 && !go) printf("  engine: no go: td_ok %d b %u in_limit %u avail %u quota %u bl1 %u run_on %u\n", (int)td_ok, st.b, in_limit, avail, st.quota, st.bl1, st.run_on);
#endif
#ifndef BROTLI_AMD_PE_OLD_RUN_REGIONS
      uint32_t want_bits = st.run_on != 0u ? PE_RUN_RBL : st.rbl;   // (a long literal run's regions take no tables per bit: four times the bits)
#ifndef BROTLI_AMD_PE_NO_CUT
      if (st.run_on == 0u && st.s_cmds != 0u && st.s_bits != 0u) {
        // A block count that runs out ends the engine's part (decode.rs:1469-1524: the switch is the checked loop's), and what the region
        // holds behind that command was built for nothing -- 3.6 regions' worth a stream of the metric's, 2.6 % of its time.  The region is
        // as long as the counts last at the rate of the region before, and an eighth.
        const float rate = (float)st.s_bits;
        float need = (float)st.bl1 * rate / (float)st.s_cmds;
        if (st.s_dsts != 0u) { const float nd = (float)st.bl2 * rate / (float)st.s_dsts; need = nd < need ? nd : need; }
        if (st.s_lits != 0u) { const float nl = (float)st.bl0 * rate / (float)st.s_lits; need = nl < need ? nl : need; }
        need = need * 1.125f + 1024.0f;
        if (need < (float)want_bits) { const uint32_t nb_ = ((uint32_t)need + 31u) & ~31u; want_bits = nb_ < PE_MIN_INPUT ? PE_MIN_INPUT : nb_; }
      }
#endif
#else
      const uint32_t want_bits = st.rbl;
#endif
      setup_tables(lbdw_, st.b & 31u, avail < want_bits ? avail : want_bits, st.run_on, st.b & 31u);
      setup_walk(st.P);
    }
    PE_BAR();   // (the region before's stores: waited for in front of the execute, which is the first to read them -- see the resolve's last barrier)
    if (pe_ctl_ld(pb, PEC_GO) == 0u) break;
    P0 = (uint64_t)pe_ctl_ld(pb, PEC_P0_LO) | ((uint64_t)pe_ctl_ld(pb, PEC_P0_HI) << 32);
#ifdef BROTLI_AMD_PROFILE_REGIONS   // (block 0: one line a region -- what it held and what it cost)
    const uint64_t rg_t0 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0 && rg_prev_end != 0) printf("   (wave 0 waited %llu ticks for the region before's last wave + set-up)\n", (unsigned long long)(rg_t0 - rg_prev_end));
#endif
    const uint32_t how = build();
#ifdef BROTLI_AMD_PROFILE_REGIONS
    const uint64_t rg_t1 = __builtin_amdgcn_s_memtime();
    if (how != 0u && blockIdx.x == 0 && threadIdx.x == 0) printf("region (literal run, how %u): %llu ticks: input+table %llu first decode %llu settle %llu ranks %llu emit+write %llu; literals %u bits %u\n", how, (unsigned long long)(rg_t1 - rg_t0),
        (unsigned long long)(rg_ts[6] - rg_t0), (unsigned long long)(rg_ts[7] - rg_ts[6]), (unsigned long long)(rg_ts[8] - rg_ts[7]), (unsigned long long)(rg_ts[9] - rg_ts[8]), (unsigned long long)(rg_t1 - rg_ts[9]), pe_ctl_ld(pb, PEC_TAKE), c.L);
    if (how != 0u && blockIdx.x == 0 && threadIdx.x == 0) printf("      rounds (parts @ tick behind the first pass): %u @ %llu, %u @ %llu, %u @ %llu, %u @ %llu\n", rg_rn[0], (unsigned long long)rg_rt[0], rg_rn[1], (unsigned long long)rg_rt[1], rg_rn[2], (unsigned long long)rg_rt[2], rg_rn[3], (unsigned long long)rg_rt[3]);
#endif
    if (how == 1u) continue;
    if (how == 2u) break;
    consume();
#ifdef BROTLI_AMD_PROFILE_REGIONS
    if (blockIdx.x == 0 && threadIdx.x == 0)
      printf("region: bits %u path %u closure %u listed %u executed %u out %u big %u dep %u staged %u | build %llu consume %llu\n", c.L, c.Rn, wn, pe_ctl_ld(pb, PEC_M), pe_ctl_ld(pb, PEC_KP),
             pe_ctl_ld(pb, PEC_OUTTOT), pe_ctl_ld(pb, PEC_NBIG), pe_ctl_ld(pb, PEC_ANYDEP), pe_ctl_ld(pb, PEC_STAGED), (unsigned long long)(rg_t1 - rg_t0), (unsigned long long)(__builtin_amdgcn_s_memtime() - rg_t1));
    if (blockIdx.x == 0 && threadIdx.x == 0)
      printf("   consume: walk+details %llu resolve %llu exec(a) %llu (b) %llu classify+wait %llu ready %llu rest(wave 0) %llu\n", (unsigned long long)(rg_ts[0] - rg_t1), (unsigned long long)(rg_ts[1] - rg_ts[0]), (unsigned long long)(rg_ts[2] - rg_ts[1]),
             (unsigned long long)(rg_ts[3] - rg_ts[2]), (unsigned long long)(rg_ts[4] > rg_ts[3] ? rg_ts[4] - rg_ts[3] : 0), (unsigned long long)(rg_ts[5] > rg_ts[4] ? rg_ts[5] - rg_ts[4] : 0), (unsigned long long)(__builtin_amdgcn_s_memtime() - (rg_ts[5] > rg_ts[3] ? rg_ts[5] : rg_ts[3])));
    rg_prev_end = __builtin_amdgcn_s_memtime(); rg_ts[4] = rg_ts[5] = 0;
#endif
    if (pe_ctl_ld(pb, PEC_CONT) == 0u) break;
  }
#elif PE_CFG_REMOTE
  // ---- a gang of blocks: the stream's regions in turns, as the two engines below, but as many regions ahead as the gang has blocks.  Where
  // the windows lie is a PLAN everybody follows without asking: region k's starts (k - first) strides behind the plan's first bit (a stride is a
  // region less a margin: a region's walk ends a command or two short of its window's end wherever it entered it).  The engine whose turn it
  // is finds out whether the stream did enter its window; if not (the region before was cut short: its path's ranks or the closure's room
  // ran out, a record hit a cap) it builds its tables once more where the stream is and writes a new plan from there -- the engines behind it,
  // who look at the plan while they wait for the stream, build theirs once more too, side by side.
  {
    // (the plan's word: generation << 48 | halvings << 44 | first region << 32 | its first bit.  Halvings: where a region's closure fills its room --
    // a stretch of few literals --, the regions from there on take half the bits, and half again if need be, as the one-block form's do; back to
    // twice the bits where the closure has become small)
    // (wave 0) region kseq's window by the plan: its tables' set-up, whether there is anything to build, the plan's generation
    auto window_by_plan = [&](const uint64_t pl) {
      // (bits 44, 45: the halvings; bits 46, 47: the stride in eighths less of a region less the margin -- where regions end short of their
      // windows' ends, as the seed of the metric's streams does, whose path's ranks run out: 9 new plans a 4 MiB stream before, a region's
      // tables of every block each)
      const uint32_t k0_ = (uint32_t)(pl >> 32) & 0xFFFu, base_ = (uint32_t)pl, shsc_ = (uint32_t)(pl >> 44) & 15u, sh_ = shsc_ & 3u;
      const uint32_t rbl_ = PE_RBL >> sh_, stride_ = ((rbl_ - PE_REMOTE_MARGIN) * (8u - (shsc_ >> 2))) >> 3;
      const uint64_t wb64 = (uint64_t)base_ + (kseq > k0_ ? (uint64_t)(kseq - k0_) * stride_ : 0ull);
      const uint32_t W = (uint32_t)((wb64 < (uint64_t)in_limit ? wb64 : (uint64_t)in_limit) >> 5);
      const uint32_t avail = (W << 5) < in_limit ? in_limit - (W << 5) : 0u;
      const uint64_t sw_ = gang_ld64(gc, GC_STOP);
      const bool buildable = td_ok && avail >= PE_MIN_INPUT && !((uint32_t)(sw_ >> 32) == epoch && (uint32_t)sw_ <= kseq);
      setup_tables(W, 0u, avail < rbl_ ? avail : rbl_, 0u, 0u);
      // (entry seeds: a region's walk ends a command or two short of bit L - 128 of its window, and the next window starts a stride behind this one's
      // start: the stream enters it that far short of rbl - stride -- PE_SEEDS bits around that.  The plan's first region starts where the stream is)
      if (kseq > k0_) { const uint32_t e0_ = rbl_ - stride_; seed_entries(e0_ > PE_SEED_BACK ? e0_ - PE_SEED_BACK : 0u, PE_SEEDS); }
      else seed_entries(0u, 32u);
      pe_ctl_st(pb, PEC_GO, buildable ? 1u : 0u); pe_ctl_st(pb, PEC_MYGEN, (uint32_t)(pl >> 48)); pe_ctl_st(pb, PEC_MYSHIFT, shsc_);
    };
    const bool alone = role == 0u && pe_ctl_ld(pb, PEC_PLAN) == 6u;   // (the owner has kept the invocation to itself: see `long_first`)
    if (!alone) for (kseq = role;; kseq += gang_m) {
      if (me == 0) window_by_plan(gang_ld64(gc, GC_PLAN));
      PE_BAR();
      PE_PROF(15);   // (the window)
      uint32_t plan;
      for (;;) {
        const bool built = pe_ctl_ld(pb, PEC_GO) != 0u;
        if (built) { const uint64_t tb_ = __builtin_amdgcn_s_memtime(); (void)tb_; (void)build(); GT(7); if (me == 0) { GANG_STAT(gc, 4, 1); if (role == 0u) GANG_STAT(gc, 20, __builtin_amdgcn_s_memtime() - tb_); } }
        // -- the stream arrives (or the plan has changed, or the invocation is over) --
        if (me == 0) {
          const uint32_t mygen = pe_ctl_ld(pb, PEC_MYGEN), want = (epoch << 12) | kseq;
          uint64_t v; uint32_t spins = 0; bool arrived, stopped, replanned;
          const uint64_t t0_ = __builtin_amdgcn_s_memtime(); (void)t0_;
          for (;;) {
            v = gang_ld64(gc, lane == 0u ? GC_ENTRY : lane == 32u ? GC_STOP : GC_PLAN);
            arrived = rdlane((uint32_t)(v >> 32), 0) == want;
            stopped = rdlane((uint32_t)(v >> 32), 32) == epoch && rdlane((uint32_t)v, 32) <= kseq;
            replanned = (rdlane((uint32_t)(v >> 32), 33) >> 16) != mygen;
            if (arrived || stopped || replanned) break;
            __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins);
          }
          GANG_STAT(gc, role == 0u ? 6 : 7, __builtin_amdgcn_s_memtime() - t0_); gs_arr = __builtin_amdgcn_s_memtime(); GT(0);

          plan = 2u;   // 0: the tables are the ones, 1: once more where the stream is, 2: the invocation is over, 3: once more by the new plan, then wait again
          if (stopped) { }
          else if (replanned) { GANG_STAT(gc, 3, 1); window_by_plan((uint64_t)rdlane((uint32_t)v, 33) | ((uint64_t)rdlane((uint32_t)(v >> 32), 33) << 32)); plan = 3u; }
          else {
            // (the bit the stream enters the region at: all the walk and the details ask for; its state comes behind the details -- see consume)
            const uint32_t eb = rdlane((uint32_t)v, 0);
            const uint32_t avail = eb < in_limit ? in_limit - ((eb >> 5) << 5) : 0u;
            const bool go = td_ok && eb < in_limit && avail >= PE_MIN_INPUT;
            if (go) {
              const uint32_t w0 = pe_ctl_ld(pb, PEC_LBDW) << 5, wl = pe_ctl_ld(pb, PEC_L), shsc_ = pe_ctl_ld(pb, PEC_MYSHIFT), sh_ = shsc_ & 3u, rbl_ = PE_RBL >> sh_;
              const bool usable = built && eb >= w0 && eb + (PE_PIPE_USEFUL >> sh_) <= w0 + wl;
              plan = usable ? 0u : 1u;
              GANG_STAT(gc, 1, 1);
              if (!usable) {
                GANG_STAT(gc, 2, 1);
                if (!built) GANG_STAT(gc, 10, 1); else if (eb < w0) GANG_STAT(gc, 11, 1); else GANG_STAT(gc, 12, 1);
                setup_tables(eb >> 5, 0u, avail < rbl_ ? avail : rbl_, 0u, 0u);
                seed_entries(0u, 32u);   // (the window starts at the entry's dword)
                // (short of the window: the regions from here on a shorter stride -- what the region before did advance, in eighths)
                uint32_t sc_ = shsc_ >> 2;
                if (built && eb < w0) {
                  const uint32_t full_ = rbl_ - PE_REMOTE_MARGIN, cur_ = (full_ * (8u - sc_)) >> 3, short_ = w0 - eb, adv_ = cur_ > short_ + 256u ? cur_ - short_ - 256u : 0u;
                  while (sc_ < 3u && ((full_ * (8u - sc_)) >> 3) > adv_) sc_++;
                }
                const uint32_t nsh_ = sh_ | (sc_ << 2);
                pe_ctl_st(pb, PEC_MYSHIFT, nsh_);
                if (lane == 0u) gang_st64(gc, GC_PLAN, ((uint64_t)((mygen + 1u) & 0xFFFFu) << 48) | ((uint64_t)nsh_ << 44) | ((uint64_t)kseq << 32) | (uint64_t)eb);
                pe_ctl_st(pb, PEC_MYGEN, (mygen + 1u) & 0xFFFFu);
                gang_drain();
              }
            }
            pe_ctl_st(pb, PEC_MYENTRY, eb);
            if (plan == 2u) plan = 4u;   // (the region cannot be taken: the invocation ends in front of it -- said once the region before is resolved, see full_arrival)
            if (plan == 0u) {
              // (the rule: the tables are the ones and their closure has room -- the walk's start words at once, one barrier between the stream's
              // arrival and the walk instead of three: the walk is what the next region waits for)
              const uint32_t shsc_ = pe_ctl_ld(pb, PEC_MYSHIFT), sh_ = shsc_ & 3u;
              if (!(wn + 64u > PE_WCAP && sh_ < 2u && pe_ctl_ld(pb, PEC_L) > (PE_RBL >> (sh_ + 1u)))) {
                pe_ctl_st(pb, PEC_LE, eb - (pe_ctl_ld(pb, PEC_LBDW) << 5));
                pe_ctl_st(pb, PEC_MYNEXT, eb);
                setup_walk(0ull);
                plan = 7u;
              }
            }
          }
          pe_ctl_st(pb, PEC_PLAN, plan);
        }
        PE_BAR();
        plan = pe_ctl_ld(pb, PEC_PLAN);
        if (plan != 3u) break;
      }
      PE_PROF(16);   // (waiting for the stream)
      if (plan == 2u) break;
      if (plan == 4u) { if (me == 0) full_arrival(false); PE_BAR(); break; }
      if (plan != 7u) {
      if (plan == 1u) (void)build();
      // a region whose closure has filled its room: half the bits, for this one and the ones behind it -- a new plan; twice, if need be
      for (uint32_t halvings = 0; halvings < 2u; halvings++) {
        if (me == 0) {
          const uint32_t shsc_ = pe_ctl_ld(pb, PEC_MYSHIFT), sh_ = shsc_ & 3u, eb = pe_ctl_ld(pb, PEC_MYENTRY);
          const uint32_t avail = eb < in_limit ? in_limit - ((eb >> 5) << 5) : 0u;
          uint32_t again = 0u;
          if (wn + 64u > PE_WCAP && sh_ < 2u && pe_ctl_ld(pb, PEC_L) > (PE_RBL >> (sh_ + 1u))) {
            const uint32_t rbl_ = PE_RBL >> (sh_ + 1u), g_ = (pe_ctl_ld(pb, PEC_MYGEN) + 1u) & 0xFFFFu;
            setup_tables(eb >> 5, 0u, avail < rbl_ ? avail : rbl_, 0u, 0u);
            seed_entries(0u, 32u);
            if (lane == 0u) gang_st64(gc, GC_PLAN, ((uint64_t)g_ << 48) | ((uint64_t)(shsc_ + 1u) << 44) | ((uint64_t)kseq << 32) | (uint64_t)eb);
            gang_drain();
            pe_ctl_st(pb, PEC_MYGEN, g_); pe_ctl_st(pb, PEC_MYSHIFT, shsc_ + 1u);
            again = 1u;
            GANG_STAT(gc, 12, 1);
          }
          pe_ctl_st(pb, PEC_PLAN, again != 0u ? 5u : 0u);
        }
        PE_BAR();
        if (pe_ctl_ld(pb, PEC_PLAN) != 5u) break;
        (void)build();
      }
      if (me == 0) {
        const uint32_t eb = pe_ctl_ld(pb, PEC_MYENTRY);
        pe_ctl_st(pb, PEC_LE, eb - (pe_ctl_ld(pb, PEC_LBDW) << 5));
        pe_ctl_st(pb, PEC_MYNEXT, eb);
        setup_walk(0ull);   // (where the region's output starts: with the stream's state, behind the details)
      }
      PE_BAR();
      }
      le = pe_ctl_ld(pb, PEC_LE);
      { const uint64_t tc_ = __builtin_amdgcn_s_memtime(); (void)tc_;
        consume();
        if (me == 0 && role == 0u) GANG_STAT(gc, 19, __builtin_amdgcn_s_memtime() - tc_); }
      if (pe_ctl_ld(pb, PEC_CONT) == 0u) break;
    }
  }
#else
  // ---- two engines: the stream's regions in turns.  An engine builds the tables of its next region while the other one takes
  // the stream through its own: the tables do not depend on where the stream enters the region (any chain of literal code
  // words is a path: they re-synchronise; the records are per state), only the walk does -- it waits for the region before's
  // resolve (PEC_RESOLVED), and the execute for the region before's output (PEC_EXECUTED).  Where a region's tables start is a
  // guess: the engine's own last region, carried on by what the stream advanced in it, less a margin.  A guess the stream
  // does not enter (it ended short of it, or too close to its end) costs the tables once more, built where the stream is.
  {
    const uint32_t entry0 = pe_ctl_ld(pbs, SCC_ENTRY);
    for (kseq = eng;; kseq += 2u) {
      // -- the window --
      if (me == 0) {
        uint32_t wbit = entry0;
        if (kseq != 0u) {
          // behind the region before's window, less the margin: a region's walk ends a command or two short of its window's end
          // wherever it entered it (that window is final once its engine has seen the stream arrive: PEC_NFINAL)
          uint32_t spins = 0;
          while (pe_ctl_ld(pbs, PEC_NFINAL) < kseq && pe_ctl_ld(pbs, PEC_STOP) == 0u) { __builtin_amdgcn_s_sleep(2); PE_SPIN_CHECK(spins); }
          lds_sync();
          wbit = (pe_ctl_ld(pbs, PEC_WINF + ((kseq - 1u) & 1u)) << 5) + PE_RBL - PE_PIPE_MARGIN;
        } else { pe_ctl_st(pbs, PEC_WINF, entry0 >> 5); lds_sync(); pe_ctl_st(pbs, PEC_NFINAL, 1u); }   // (the first region's is where the stream is)
        const uint32_t W = wbit >> 5;
        const uint32_t avail = (W << 5) < in_limit ? in_limit - (W << 5) : 0u;
        const bool buildable = td_ok && avail >= PE_MIN_INPUT && pe_ctl_ld(pbs, PEC_STOP) == 0u;
        setup_tables(W, 0u, avail < PE_RBL ? avail : PE_RBL, 0u, 0u);
        pe_ctl_st(pb, PEC_GO, buildable ? 1u : 0u);
      }
      PE_BAR();
      PE_PROF(15);   // (waiting for the window)
      bool built = false;
      if (pe_ctl_ld(pb, PEC_GO) != 0u) { (void)build(); built = true; }
      // -- the stream arrives --
      if (me == 0) {
        uint32_t spins = 0;
        while (pe_ctl_ld(pbs, PEC_RESOLVED) < kseq && pe_ctl_ld(pbs, PEC_STOP) == 0u) { __builtin_amdgcn_s_sleep(2); PE_SPIN_CHECK(spins); }
        lds_sync();
        uint32_t plan = 2u;   // 0: the tables are the ones, 1: once more where the stream is, 2: the invocation is over
        if (pe_ctl_ld(pbs, PEC_RESOLVED) >= kseq && pe_ctl_ld(pbs, PEC_STOP) == 0u) {
          const PeStream st = pe_st_load(pbs);
          const uint32_t avail = st.b < in_limit ? in_limit - ((st.b >> 5) << 5) : 0u;
          const bool go = td_ok && st.b < in_limit && avail >= PE_MIN_INPUT && st.quota >= SC_MIN_QUOTA && st.bl1 != 0u;
          if (go) {
            const uint32_t w0 = pe_ctl_ld(pb, PEC_LBDW) << 5, wl = pe_ctl_ld(pb, PEC_L);
            const bool usable = built && st.b >= w0 && st.b + PE_PIPE_USEFUL <= w0 + wl;
            plan = usable ? 0u : 1u;
            if (!usable) setup_tables(st.b >> 5, 0u, avail < PE_RBL ? avail : PE_RBL, 0u, 0u);
            if (kseq != 0u) { pe_ctl_st(pbs, PEC_WINF + (kseq & 1u), usable ? w0 >> 5 : st.b >> 5); lds_sync(); pe_ctl_st(pbs, PEC_NFINAL, kseq + 1u); }
          }
        }
        if (plan == 2u) pe_ctl_st(pbs, PEC_STOP, 1u);
        pe_ctl_st(pb, PEC_PLAN, plan);
      }
      PE_BAR();
      PE_PROF(16);   // (waiting for the stream)
      const uint32_t plan = pe_ctl_ld(pb, PEC_PLAN);
      if (plan == 2u) break;
      if (plan == 1u) (void)build();
      if (me == 0) {
        const PeStream st = pe_st_load(pbs);
        pe_ctl_st(pb, PEC_LE, st.b - (pe_ctl_ld(pb, PEC_LBDW) << 5));
        pe_ctl_st(pb, PEC_MYENTRY, st.b); pe_ctl_st(pb, PEC_MYNEXT, st.b);
        setup_walk(st.P);
#ifdef BROTLI_AMD_PE_DEBUG
        if (blockIdx.x == 0 && lane == 0) printf("region %u (engine %u): window dword %u, %u bits, entry %u (bit %u of it), plan %u, P %llu, bl1 %u, Rn %u wn %u\n", kseq, eng, pe_ctl_ld(pb, PEC_LBDW), pe_ctl_ld(pb, PEC_L), st.b, st.b - (pe_ctl_ld(pb, PEC_LBDW) << 5), plan, (unsigned long long)st.P, st.bl1, pe_ctl_ld(pb, PEC_RN), wn);
#endif
      }
      PE_BAR();
      le = pe_ctl_ld(pb, PEC_LE);
      P0 = (uint64_t)pe_ctl_ld(pb, PEC_P0_LO) | ((uint64_t)pe_ctl_ld(pb, PEC_P0_HI) << 32);
      consume();
      if (pe_ctl_ld(pb, PEC_CONT) == 0u) break;
    }
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t seq_ = hc_ld(HC_SEQ);   // (the request this invocation answers: the decoding wave posts the next one behind the barrier)
  __syncthreads();  // every store of the engine is in memory before the decoding wave goes on alone
  if (REMOTE && role != 0u) {   // a helper block: it has left the invocation, and waits for the next one
    if (threadIdx.x == 0u) gang_add32(gc, GC_READY, 1u);
    goto pe_again;
  }
  if (rfl(me_) != 0u) {
#ifdef BROTLI_AMD_PE_NO_STAY   // (for A/B: every wave returns after every invocation, as in round 3)
    return seq_;
#endif
    for (uint32_t idle = 0;; idle++) {   // (as helper_wave idles)
      if (hc_ld(HC_SEQ) != seq_) break;
      if (idle < 256u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(127);
    }
    lds_acquire();
    if (hc_ld(HC_SEQ) == seq_ + 1u && hc_ld(HC_KIND) == (REMOTE ? (uint32_t)HK_PATHR : PIPE2 ? (uint32_t)HK_PATH2 : PE_DICT ? (uint32_t)HK_PATHG : (uint32_t)HK_PATH)) goto pe_again;   // (this form of the engine again: the lean one and the general one are two functions)
    return seq_;
  }
#ifdef BROTLI_AMD_PROFILE_SCAN
  if (blockIdx.x == 0 && lane == 0) { for (int k = 0; k < 32; k++) if (k != 30) g_path_prof[k] += pp_acc[k]; g_path_prof[32] += pe_ctl_ld(pbs, PEC_STATE + 6); g_path_prof[33] += 1; }
#endif
  if (REMOTE && pe_ctl_ld(pb, PEC_PLAN) != 6u) {   // (6: the owner kept the invocation to itself)
    // (the owner of a gang) the invocation's end as the gang left it: the regions resolved in all, the stream's state behind the last of them
    // -- whoever's it was --, and the last one's output in memory
    uint32_t spins = 0; uint64_t sw_, v; (void)spins;
    const uint64_t t0_ = __builtin_amdgcn_s_memtime(); (void)t0_;
    for (;;) { sw_ = gang_ld64(gc, GC_STOP); if ((uint32_t)(sw_ >> 32) == epoch) break; __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins); }
    const uint32_t Kr = (uint32_t)sw_, want = (epoch << 12) | Kr;
    for (;;) {
      v = gang_ld64(gc, GC_STATE + 8u * (lane < GC_STATE_WORDS ? lane : 0u));
      if (__ballot(lane < GC_STATE_WORDS && (uint32_t)(v >> 32) == want) == ((1ull << GC_STATE_WORDS) - 1ull)) break;
      __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins);
    }
    if (lane < 25u) lds_st32(pbs + PE_CTL + 4u * (PEC_STATE + lane), (uint32_t)v);
    pe_ctl_st(pbs, PEC_DECLINE, (rdlane((uint32_t)v, 25) >> 1) & 3u);
    pe_ctl_st(pb, PEC_DSEEN, Kr != 0u ? (rdlane((uint32_t)v, 25) >> 3) & 1u : 0u);   // (the engine's part ended in front of a dictionary reference: the general form's stream, as the lean form says it)
    if (Kr != 0u) {
      for (;;) { const uint64_t ew = gang_ld64(gc, GC_EXEC); if ((uint32_t)(ew >> 32) == epoch && (uint32_t)ew >= Kr) break; __builtin_amdgcn_s_sleep(1); PE_SPIN_CHECK(spins); }
      gang_acquire();
    }
    GANG_STAT(gc, 31, Kr == 0u ? 1u : 0u);
    GANG_STAT(gc, 13, __builtin_amdgcn_s_memtime() - t0_); GANG_STAT(gc, 16, __builtin_amdgcn_s_memtime() - gs_t0); GANG_STAT(gc, 14, Kr); GANG_STAT(gc, 15, rdlane((uint32_t)v, 25) >> 1 & 1u);
    lds_sync();
  }
  // ---- hand the stream back in front of the next command (LDS_LEAN, as the scan engine does) ----
  const PeStream st_ = pe_st_load(pbs);
  PeStream st = st_;
  if (REMOTE) { GANG_STAT(gc, 21, st.ncmd < 64u ? 1u : 0u); GANG_STAT(gc, 22, st.ncmd); GANG_STAT(gc, 23, st.ncmd == 0u ? 1u : 0u); }
  if (st.run_on != 0u) {
    // inside a command whose literal run had regions of its own: the checked loop finishes its literals (what the limits kept
    // back, or none), its distance and its copy; the reference takes a command's whole insert length off when it reads the
    // command (decode.rs:2388)
    st.mlen -= (int32_t)st.run_rem;
    if (lane == 0) {
      LEAN_ST(L_SC_POS_LO, st.b); LEAN_ST(L_SC_POS_HI, (uint32_t)SCX_LITERALS_REST);
      LEAN_ST(L_P_LO, (uint32_t)st.P); LEAN_ST(L_P_HI, (uint32_t)(st.P >> 32)); LEAN_ST(L_QUOTA, st.quota); LEAN_ST(L_MLEN, st.mlen);
      LEAN_ST(L_BL0, st.bl0); LEAN_ST(L_BL1, st.bl1); LEAN_ST(L_BL2, st.bl2);
      LEAN_ST(L_D0, st.d0); LEAN_ST(L_D1, st.d1); LEAN_ST(L_D2, st.d2); LEAN_ST(L_D3, st.d3); LEAN_ST(L_NCMD_LO, st.ncmd);
      LEAN_ST(L_INSERT, st.run_rem); LEAN_ST(L_COPY, st.run_copy); LEAN_ST(L_DCODE, st.run_implicit ? 0 : -1); LEAN_ST(L_DCTX, st.run_dctx); LEAN_ST(L_LITS_LEFT, st.run_rem);
    }
    lds_sync();
    return st.ncmd;
  }
  const bool pdx = PE_DICT && pe_ctl_ld(pb, PEC_PDX) != 0u;   // (behind the distance of a command whose literals are out: postReadDistance, decode.rs:2583)
  if (lane == 0) {
    LEAN_ST(L_SC_POS_LO, st.b); LEAN_ST(L_SC_POS_HI, pdx ? (uint32_t)SCX_POST_DISTANCE : (uint32_t)SCX_BEGIN | (PIPE && (pe_ctl_ld(pbs, PEC_DECLINE) & 1u) != 0u ? 0x100u : 0u) | (REMOTE && (pe_ctl_ld(pbs, PEC_DECLINE) & 2u) != 0u ? 0x800u : 0u) | (REMOTE && pe_ctl_ld(pb, PEC_PLAN) == 6u && pe_ctl_ld(pb, PEC_NOHELP) != 0u ? 0x1000u : 0u) | (!PIPE && pe_ctl_ld(pb, PEC_OVF) >= 3u ? 0x200u : 0u) | (!PIPE2 && !PE_DICT && pe_ctl_ld(pb, PEC_DSEEN) != 0u ? 0x400u : 0u));
    LEAN_ST(L_P_LO, (uint32_t)st.P); LEAN_ST(L_P_HI, (uint32_t)(st.P >> 32)); LEAN_ST(L_QUOTA, st.quota); LEAN_ST(L_MLEN, st.mlen);
    LEAN_ST(L_BL0, st.bl0); LEAN_ST(L_BL1, st.bl1); LEAN_ST(L_BL2, st.bl2);
    LEAN_ST(L_D0, st.d0); LEAN_ST(L_D1, st.d1); LEAN_ST(L_D2, st.d2); LEAN_ST(L_D3, st.d3); LEAN_ST(L_NCMD_LO, st.ncmd);
    LEAN_ST(L_INSERT, 0u); LEAN_ST(L_COPY, pdx ? pe_ctl_ld(pb, PEC_DICTN) : 0u); LEAN_ST(L_DCODE, pdx ? (int32_t)pe_ctl_ld(pb, PEC_DICTD) : 0); LEAN_ST(L_DCTX, 0u); LEAN_ST(L_LITS_LEFT, 0u);
  }
  lds_sync();
  return st.ncmd;
}
}  // namespace PE_CFG_NS
#undef PE_PROF
#undef PE_COUNT
#undef PE_LANECOUNT
#undef PE_TRY_RUN
#undef PE_TRY_RUN_FROM
#undef PE_BAR
#undef PE_HOPS_REC
#undef PE_DICT
#undef PE_SPIN_CHECK
#undef RG_STAMP
#undef GT
#undef GTC
