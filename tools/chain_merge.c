/* chain_merge -- how fast does a Brotli command parse that was started at a WRONG bit fall in with the true chain?
 *
 * Test/analysis tool.  Links oracle/brotli_oracle.c built with -DORACLE_STATS: the oracle reports
 * every true command (bit position of its head) and every block switch; at each metablock start and behind each block
 * switch this tool starts probe parses (brotli_oracle_probe_chain: a command is assumed to begin at the probe's first bit,
 * current block types, infinite block counts) every STRIDE bits over the next SPAN bits.  After the stream has been
 * decoded, each probe is followed until its first command start that is also a true command start -- from there on it IS
 * the true chain (same trees, same position).  Only the part of a probe inside its own segment (up to the next block
 * switch / metablock end) is judged.  Prints histograms of commands / bits / symbols until the merge.
 *
 * usage: chain_merge <file.br> [stride=509] [max_cmds=2000]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef struct { int32_t result, error_code; uint64_t decoded_size, consumed, produced; uint32_t window_bits, num_metablocks; uint64_t num_commands, num_literals, num_context_literals; uint32_t max_literal_trees, max_block_types; } OracleInfo;
int brotli_oracle_decode(const uint8_t* in, size_t in_size, uint8_t* out, size_t out_cap, uint32_t flags, OracleInfo* info);
int brotli_oracle_probe_chain(const void* state, uint64_t start, uint64_t end_bit, uint64_t* cmd_pos, uint32_t* cmd_syms, int max_cmds);

static uint64_t g_total_bits, g_stride = 509; static int g_max_cmds = 2000;
static uint8_t* g_true;      /* bitmap of true command starts */
static uint64_t* g_true_list; static size_t g_ntrue;
typedef struct { uint64_t start, seg_begin; uint32_t first, n; } Probe;   /* slice of g_pos / g_syms */
static Probe* g_probes; static size_t g_nprobes, g_cap_probes;
static uint64_t* g_pos; static uint32_t* g_syms; static size_t g_npos, g_cap_pos;
static uint64_t* g_seg; static size_t g_nseg;   /* segment starts (bit positions), increasing */
static uint64_t g_switches[3];
static uint8_t* g_path; static const void* g_state; static uint64_t g_region = 32768, g_region_end;
static uint64_t g_unsync_hist[66], g_unsync_cmds, g_onpath0, g_unsync_capped;
void brotli_oracle_literal_path(const void* state, uint64_t start, uint64_t end_bit, uint8_t* bitmap);
uint32_t brotli_oracle_literal_len(const void* state, uint64_t pos);

static void start_probes(uint64_t from, const void* state) {
  g_seg = realloc(g_seg, (g_nseg + 1) * sizeof *g_seg); g_seg[g_nseg++] = from;
  uint64_t span = 600000; /* longer than any block of the bench streams; probes are cut at the segment's end afterwards */
  for (uint64_t p = from + g_stride; p < from + span && p < g_total_bits; p += g_stride) {
    if (g_npos + (size_t)g_max_cmds > g_cap_pos) { g_cap_pos = g_cap_pos * 2 + (1u << 20); g_pos = realloc(g_pos, g_cap_pos * 8); g_syms = realloc(g_syms, g_cap_pos * 4); }
    if (g_nprobes == g_cap_probes) { g_cap_probes = g_cap_probes * 2 + 1024; g_probes = realloc(g_probes, g_cap_probes * sizeof *g_probes); }
    int n = brotli_oracle_probe_chain(state, p, g_total_bits, g_pos + g_npos, g_syms + g_npos, g_max_cmds);
    g_probes[g_nprobes++] = (Probe){p, from, (uint32_t)g_npos, (uint32_t)n};
    g_npos += (size_t)n;
  }
}
void brotli_oracle_describe_metablock(const void* state, FILE* f);
void oracle_stats_metablock(uint64_t first_bit, const void* state) { g_state = state; g_region_end = 0; brotli_oracle_describe_metablock(state, stdout); start_probes(first_bit, state); }
void oracle_stats_switch(int category, uint64_t bit, uint64_t resume_bit, const void* state) { (void)bit; g_switches[category]++; start_probes(resume_bit, state); }
void oracle_stats_cmd(uint64_t cmd_pos, uint64_t lit_pos, uint64_t end_pos, int32_t ins, int32_t copy, uint32_t dsym, int32_t dist, uint64_t P) {
  (void)end_pos; (void)copy; (void)dsym; (void)dist; (void)P;
  /* literal path of the region this command starts (regions begin at a command boundary, as the engine's would) */
  if (cmd_pos >= g_region_end) { g_region_end = cmd_pos + g_region; brotli_oracle_literal_path(g_state, cmd_pos, g_region_end + 64 < g_total_bits ? g_region_end + 64 : g_total_bits, g_path); }
  if (ins > 0) { /* literals decoded from the run's first bit until the chain is on the path */
    uint64_t x = lit_pos; int k = 0;
    while (k < ins && k < 64 && !(g_path[x >> 3] >> (x & 7) & 1)) { x += brotli_oracle_literal_len(g_state, x); k++; }
    g_unsync_hist[k]++; g_unsync_cmds++; if (k == 0) g_onpath0++; if (k == ins || k == 64) g_unsync_capped++;
  }
  g_true[cmd_pos >> 3] |= (uint8_t)(1u << (cmd_pos & 7));
  g_true_list[g_ntrue++] = cmd_pos;
}
static int cmp64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static void pct(const char* what, uint64_t* v, size_t n) {
  qsort(v, n, 8, cmp64);
  double mean = 0; for (size_t i = 0; i < n; i++) mean += (double)v[i]; mean /= (double)(n ? n : 1);
  printf("  %-22s mean %8.1f  p10 %6llu  p50 %6llu  p75 %6llu  p90 %6llu  p99 %6llu  max %7llu\n", what, mean,
         (unsigned long long)v[n / 10], (unsigned long long)v[n / 2], (unsigned long long)v[n * 3 / 4], (unsigned long long)v[n * 9 / 10], (unsigned long long)v[n * 99 / 100], (unsigned long long)v[n - 1]);
}
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: chain_merge <file.br> [stride] [max_cmds]\n"); return 2; }
  if (argc > 2) g_stride = strtoull(argv[2], 0, 10);
  if (argc > 3) g_max_cmds = atoi(argv[3]);
  FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* in = malloc((size_t)n + 8); if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2; fclose(f);
  g_total_bits = (uint64_t)n * 8; g_true = calloc((size_t)n + 8, 1); g_path = calloc((size_t)n + 64, 1); g_true_list = malloc(((size_t)n + 8) * 8);
  size_t cap = 256u << 20; uint8_t* out = malloc(cap);
  OracleInfo info; memset(&info, 0, sizeof info);
  brotli_oracle_decode(in, (size_t)n, out, cap, 1, &info);
  printf("stream: %ld bytes -> %llu, %u metablocks, %llu commands, %llu literals; block switches literal/command/distance %llu/%llu/%llu; %zu segments\n", n,
         (unsigned long long)info.decoded_size, info.num_metablocks, (unsigned long long)info.num_commands, (unsigned long long)info.num_literals,
         (unsigned long long)g_switches[0], (unsigned long long)g_switches[1], (unsigned long long)g_switches[2], g_nseg);
  /* judge the probes */
  uint64_t* mc = malloc(g_nprobes * 8), *mb = malloc(g_nprobes * 8), *ms = malloc(g_nprobes * 8); size_t nm = 0, never = 0, judged = 0;
  for (size_t i = 0; i < g_nprobes; i++) {
    Probe* p = &g_probes[i];
    /* segment end = next segment start after seg_begin */
    uint64_t seg_end = g_total_bits;
    for (size_t k = 0; k < g_nseg; k++) if (g_seg[k] > p->seg_begin) { seg_end = g_seg[k]; break; }
    if (p->start >= seg_end) continue; /* started behind the next switch: that switch's own probes cover it */
    judged++;
    int merged = 0;
    for (uint32_t k = 0; k < p->n; k++) {
      uint64_t q = g_pos[p->first + k];
      if (q >= seg_end) break;
      if (g_true[q >> 3] >> (q & 7) & 1) { mc[nm] = k; mb[nm] = q - p->start; ms[nm] = g_syms[p->first + k]; nm++; merged = 1; break; }
    }
    if (!merged) never++;
  }
  printf("probes every %llu bits: %zu judged, %zu merged inside their segment, %zu did not (segment or %d-command budget ended first)\n",
         (unsigned long long)g_stride, judged, nm, never, g_max_cmds);
  if (nm) { pct("commands until merge", mc, nm); pct("bits until merge", mb, nm); pct("symbols until merge", ms, nm); }
  printf("literal path (regions of %llu bits): %llu commands with literals; first literal on the path %.1f %%; run ends (or 64) before the path is met %.1f %%\n  literals until on path:", (unsigned long long)g_region, (unsigned long long)g_unsync_cmds, 100.0 * (double)g_onpath0 / (double)g_unsync_cmds, 100.0 * (double)g_unsync_capped / (double)g_unsync_cmds);
  { uint64_t acc = 0; for (int k = 0; k <= 64; k++) { acc += g_unsync_hist[k]; if (k <= 12 || k == 16 || k == 24 || k == 32 || k == 64) printf(" <=%d: %.1f%%", k, 100.0 * (double)acc / (double)g_unsync_cmds); } printf("\n"); }
  /* true chain, for scale */
  if (g_ntrue > 1) printf("true chain: %.1f bits per command\n", (double)(g_true_list[g_ntrue - 1] - g_true_list[0]) / (double)(g_ntrue - 1));
  return 0;
}
