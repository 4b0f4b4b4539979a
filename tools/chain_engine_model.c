/* chain_engine_model -- CPU model of the chain engine's parse phase (round 4), to size it before it is written in HIP.
 *
 * Test/analysis tool.  Links oracle/brotli_oracle.c built with -DORACLE_STATS.  The chain engine parses one stream with
 * one CHAIN per lane: chain k starts at bit entry + k * DELTA as if a command began there and parses command after
 * command (ReadCommandInternal, the literals skipped, ReadDistanceInternal: src/decode.rs:2134-2189, 2393-2462,
 * 2066-2131), all lanes of a wave in lockstep, one command a step.  Chain 0 starts at the stream's real position.  A chain
 * stops when the command start it arrives at is one the chain that OWNS that part of the stream (the chain that started in
 * it) has already visited: from there on the two are the same.  The true command list is chain 0's up to its link, then the
 * linked chain's from there, and so on.  This model replays that in lockstep on the true streams and reports what it costs:
 * wave steps, literal sub-steps (a step takes as long as its lane with the most literals), redundant parses, and how much
 * of the stream one round recovers.
 *
 * usage: chain_engine_model <file.br> [delta_bits=3072] [lanes=1024] [max_steps=4096] [nmax=512]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef struct { int32_t result, error_code; uint64_t decoded_size, consumed, produced; uint32_t window_bits, num_metablocks; uint64_t num_commands, num_literals, num_context_literals; uint32_t max_literal_trees, max_block_types; } OracleInfo;
int brotli_oracle_decode(const uint8_t* in, size_t in_size, uint8_t* out, size_t out_cap, uint32_t flags, OracleInfo* info);
int brotli_oracle_probe_chain2(const void* state, uint64_t start, uint64_t end_bit, uint64_t* cmd_pos, uint32_t* cmd_ins, uint8_t* cmd_dist, int max_cmds, uint64_t* next_pos);
void brotli_oracle_block_lengths(const void* state, uint32_t out[3]);

static uint64_t g_total_bits, g_delta = 3072; static int g_lanes = 1024, g_max_steps = 4096; static uint32_t g_nmax = 512;
static double g_A = 110.0, g_B = 12.0;   /* instructions of a step's command part, and per literal sub-step (estimates) */
/* totals */
static uint64_t t_true, t_rounds, t_wave_steps, t_sub_steps, t_lane_steps, t_recovered, t_invocations, t_maxsteps_sum, t_dead, t_unlinked, t_steps_crit;
static uint64_t t_byhand;
static uint64_t t_sub_k[4];   /* literal sub-steps if a step takes at most K = 8, 16, 32, 64 literals a lane (the rest next step) */

typedef struct { uint64_t* pos; uint32_t* ins; uint8_t* dist; int n; uint64_t next; int link_chain, link_idx, stop_step, cur, active, dead; } Chain;

static void simulate_invocation(const void* state, uint64_t from) {
  uint32_t bl[3]; brotli_oracle_block_lengths(state, bl);
  /* the true chain of this invocation: up to the first block count that runs out */
  int cap = 1 << 20;
  uint64_t* tp = malloc((size_t)cap * 8); uint32_t* ti = malloc((size_t)cap * 4); uint8_t* td = malloc((size_t)cap); uint64_t tnext;
  int tn = brotli_oracle_probe_chain2(state, from, g_total_bits, tp, ti, td, cap, &tnext);
  { uint64_t lits = 0, dists = 0; int k = 0;
    for (; k < tn; k++) { if ((uint32_t)k >= bl[1]) break; lits += ti[k]; dists += td[k]; if (lits > bl[0] || dists > bl[2]) break; }
    tn = k; }
  t_true += (uint64_t)tn; t_invocations++;
  int done = 0; double est_bits = 96.0;
  while (done < tn) {
    const uint64_t entry = tp[done];
    const int left = tn - done;
    uint64_t extent = (uint64_t)(est_bits * (double)left * 1.05) + 256;
    if (entry + extent > g_total_bits) extent = g_total_bits - entry;
    int N = (int)((extent + g_delta - 1) / g_delta);
    if (N > g_lanes) { N = g_lanes; extent = (uint64_t)N * g_delta; }
    if (N < 1) N = 1;
    Chain* c = calloc((size_t)N, sizeof *c);
    for (int k = 0; k < N; k++) {
      c[k].pos = malloc((size_t)(g_max_steps + 1) * 8); c[k].ins = malloc((size_t)(g_max_steps + 1) * 4); c[k].dist = malloc((size_t)g_max_steps + 1);
      c[k].n = brotli_oracle_probe_chain2(state, entry + (uint64_t)k * g_delta, g_total_bits, c[k].pos, c[k].ins, c[k].dist, g_max_steps, &c[k].next);
      c[k].link_chain = -1; c[k].active = 1; c[k].cur = 0; c[k].stop_step = -1;
    }
    /* lockstep: at step t every active chain parses its command number t */
    int steps = 0;
    const int waves = (N + 63) / 64;
    for (int t = 0; t < g_max_steps; t++) {
      int any = 0;
      for (int w = 0; w < waves; w++) {
        uint32_t maxn = 0; int wany = 0; uint32_t nk[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; l++) {
          const int k = w * 64 + l; if (k >= N || !c[k].active) continue;
          if (t >= c[k].n) { c[k].active = 0; c[k].stop_step = t; continue; }   /* ran out of stream */
          wany = 1; t_lane_steps++;
          const uint32_t n = c[k].ins[t];
          if (n > g_nmax) { c[k].active = 0; c[k].dead = 1; c[k].stop_step = t; c[k].n = t; t_dead++; continue; }   /* a literal run the engine does not take: the chain ends in front of it */
          if (n > maxn) maxn = n;
          c[k].cur = t + 1;
        }
        if (wany) { t_wave_steps++; t_sub_steps += maxn; any = 1;
          (void)nk; }
      }
      /* after the step: where each chain stands; does the owner of that part know the place?  (the owner's records up to this step) */
      for (int k = 0; k < N; k++) {
        if (!c[k].active) continue;
        const uint64_t p = c[k].cur < c[k].n ? c[k].pos[c[k].cur] : c[k].next;
        if (p >= entry + extent) { c[k].active = 0; c[k].stop_step = t + 1; c[k].n = c[k].cur; continue; }   /* beyond the round's part of the stream */
        const int j = (int)((p - entry) / g_delta);
        if (j <= k) continue;
        /* binary search in chain j's positions [0, cur_j] */
        int lo = 0, hi = c[j].cur < c[j].n ? c[j].cur : c[j].n - 1;
        if (c[j].dead && hi >= c[j].n) hi = c[j].n - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[j].pos[mid] < p) lo = mid + 1; else hi = mid; }
        if (lo < c[j].n && lo <= c[j].cur && c[j].pos[lo] == p) { c[k].active = 0; c[k].link_chain = j; c[k].link_idx = lo; c[k].stop_step = t + 1; c[k].n = c[k].cur; }
      }
      if (!any) break;
      steps = t + 1;
    }
    t_maxsteps_sum += (uint64_t)steps;
    /* stitch: chain 0, then its link, ... */
    int k = 0, idx = 0, got = 0, ok = 1; int crit = 0;
    for (;;) {
      for (int i = idx; i < c[k].n; i++) {
        if (done + got >= tn) break;
        if (c[k].pos[i] != tp[done + got]) { ok = 0; break; }
        got++;
      }
      if (c[k].stop_step > crit) crit = c[k].stop_step;
      if (!ok || done + got >= tn || c[k].link_chain < 0) break;
      idx = c[k].link_idx; k = c[k].link_chain;
    }
    if (!ok) { fprintf(stderr, "MISMATCH: stitched chain leaves the true chain (round at bit %llu)\n", (unsigned long long)entry); exit(1); }
    if (c[k].link_chain < 0 && done + got < tn) t_unlinked++;
    t_steps_crit += (uint64_t)crit;
    if (got == 0) {   /* the round's first command is one the engine does not take (a long literal run): the checked loop's */
      if (ti[done] > g_nmax) { t_byhand++; done++; for (int q = 0; q < N; q++) { free(c[q].pos); free(c[q].ins); free(c[q].dist); } free(c); continue; }
      fprintf(stderr, "no progress at bit %llu\n", (unsigned long long)entry); exit(1);
    }
    if (got > 8) est_bits = (double)((done + got < tn ? tp[done + got] : tnext) - entry) / (double)got;
    if (getenv("CEM_VERBOSE")) printf("round at bit %llu: %d left, extent %llu, %d chains, %d steps, got %d, ended on chain %d (n %d, link %d, dead %d, stop step %d, next pos %llu = entry + %llu)\n", (unsigned long long)entry, left, (unsigned long long)extent, N, steps, got, k, c[k].n, c[k].link_chain, c[k].dead, c[k].stop_step,
        (unsigned long long)c[k].next, (unsigned long long)(c[k].next - entry));
    t_recovered += (uint64_t)got; t_rounds++;
    done += got;
    for (int q = 0; q < N; q++) { free(c[q].pos); free(c[q].ins); free(c[q].dist); }
    free(c);
  }
  free(tp); free(ti); free(td);
}
void oracle_stats_metablock(uint64_t first_bit, const void* state) { simulate_invocation(state, first_bit); }
void oracle_stats_switch(int category, uint64_t bit, uint64_t resume_bit, const void* state) { (void)category; (void)bit; simulate_invocation(state, resume_bit); }
void oracle_stats_cmd(uint64_t a, uint64_t b, uint64_t c, int32_t d, int32_t e, uint32_t f, int32_t g, uint64_t h) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: chain_engine_model <file.br> [delta_bits] [lanes] [max_steps] [nmax]\n"); return 2; }
  if (argc > 2) g_delta = strtoull(argv[2], 0, 10);
  if (argc > 3) g_lanes = atoi(argv[3]);
  if (argc > 4) g_max_steps = atoi(argv[4]);
  if (argc > 5) g_nmax = (uint32_t)atoi(argv[5]);
  FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* in = malloc((size_t)n + 8); if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2; fclose(f);
  g_total_bits = (uint64_t)n * 8;
  size_t cap = 256u << 20; uint8_t* out = malloc(cap);
  OracleInfo info; memset(&info, 0, sizeof info);
  brotli_oracle_decode(in, (size_t)n, out, cap, 1, &info);
  printf("stream: %ld -> %llu bytes, %llu commands, %llu literals\n", n, (unsigned long long)info.decoded_size, (unsigned long long)info.num_commands, (unsigned long long)info.num_literals);
  printf("chains every %llu bits on %d lanes: %llu invocations, %llu rounds, %llu of %llu commands recovered; chains that ended in front of a long literal run %llu; rounds whose stitched chain ended unlinked %llu\n",
         (unsigned long long)g_delta, g_lanes, (unsigned long long)t_invocations, (unsigned long long)t_rounds, (unsigned long long)t_recovered, (unsigned long long)t_true, (unsigned long long)t_dead, (unsigned long long)t_unlinked);
  const double cmds = (double)(t_true ? t_true : 1);
  printf("  commands left to the checked loop %llu\n", (unsigned long long)t_byhand);
  printf("  lane steps %llu (%.2f per command), wave steps %llu, literal sub-steps %llu (%.1f per wave step); steps of the rounds together %llu, of their stitched chains %llu\n",
         (unsigned long long)t_lane_steps, (double)t_lane_steps / cmds, (unsigned long long)t_wave_steps, (unsigned long long)t_sub_steps, (double)t_sub_steps / (double)(t_wave_steps ? t_wave_steps : 1),
         (unsigned long long)t_maxsteps_sum, (unsigned long long)t_steps_crit);
  const double instr = g_A * (double)t_wave_steps + g_B * (double)t_sub_steps;
  printf("  estimate at %.0f + %.0f x literals instructions a wave step: %.0f instructions = %.1f per command\n", g_A, g_B, instr, instr / cmds);
  return 0;
}
