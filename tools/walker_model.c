/* walker_model -- CPU model of the walker engine's first pass (round 5), to size it before it is written in HIP.
 *
 * Test/analysis tool.  Links oracle/brotli_oracle.c built with -DORACLE_STATS.
 *
 * The walker engine parses a metablock with one serial WALKER per lane: a walker takes a seed (a bit position, as if a
 * command began there), parses command after command symbol by symbol (ReadCommandInternal, the literals one code word
 * after the other, ReadDistanceInternal: src/decode.rs:2134-2189, 2393-2462, 2066-2131) and marks every command start it
 * visits.  It stops at a command start somebody has marked before (from there on the two chains are one) and takes the
 * next seed.  Seed 0 is the stream's real position; the others lie `delta` bits apart.  The true command list is seed
 * 0's chain up to the mark that stopped it, then the marking chain's from there, and so on.
 *
 * A wave's loop ("tick"): lanes at a command start parse the distance in front of it and the command (A instructions if
 * any lane does), then every lane decodes up to K literals (B instructions a literal, as many as the lane with the most).
 * The model replays that on true streams and reports: walker steps per true command (redundancy), ticks until the true
 * chain is complete (depth), wave instructions per true command.
 *
 * usage: walker_model <file.br> [delta_bits=2048] [lanes=256] [K=16] [nmax=2048]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef struct { int32_t result, error_code; uint64_t decoded_size, consumed, produced; uint32_t window_bits, num_metablocks; uint64_t num_commands, num_literals, num_context_literals; uint32_t max_literal_trees, max_block_types; } OracleInfo;
int brotli_oracle_decode(const uint8_t* in, size_t in_size, uint8_t* out, size_t out_cap, uint32_t flags, OracleInfo* info);
int brotli_oracle_probe_chain2(const void* state, uint64_t start, uint64_t end_bit, uint64_t* cmd_pos, uint32_t* cmd_ins, uint8_t* cmd_dist, int max_cmds, uint64_t* next_pos);
void brotli_oracle_block_lengths(const void* state, uint32_t out[3]);

static uint64_t g_total_bits, g_delta = 2048; static int g_lanes = 256, g_K = 16; static uint32_t g_nmax = 2048;
static double g_A = 110.0, g_B = 13.0;
static uint64_t t_true, t_inv, t_ticks, t_ticks_crit, t_lane_cmd, t_lane_lit, t_wave_cmd, t_wave_lit, t_seeds, t_byhand, t_rounds;
static double t_instr;

#define CHUNK 64
typedef struct { uint64_t* pos; uint32_t* ins; uint8_t* dist; uint32_t* tick; int n, cap; uint64_t next; int link_chain, link_idx, walked; } Chain;
static void chain_more(const void* state, Chain* ch, uint64_t from, uint64_t end) {   /* CHUNK more commands of the chain */
  if (ch->n + CHUNK + 1 > ch->cap) { ch->cap = ch->cap ? ch->cap * 2 : 2 * CHUNK + 2; ch->pos = realloc(ch->pos, (size_t)ch->cap * 8); ch->ins = realloc(ch->ins, (size_t)ch->cap * 4); ch->dist = realloc(ch->dist, (size_t)ch->cap); ch->tick = realloc(ch->tick, (size_t)ch->cap * 4); }
  if (from >= end) { ch->next = from; return; }
  ch->n += brotli_oracle_probe_chain2(state, from, end, ch->pos + ch->n, ch->ins + ch->n, ch->dist + ch->n, CHUNK, &ch->next);
}

typedef struct { int chain, idx; uint32_t rem; int state; } Lane;   /* state 0 idle, 1 at a command start, 2 in literals */

static void simulate(const void* state, uint64_t from) {
  uint32_t bl[3]; brotli_oracle_block_lengths(state, bl);
  int cap = 1 << 20;
  uint64_t* tp = malloc((size_t)cap * 8); uint32_t* ti = malloc((size_t)cap * 4); uint8_t* td = malloc((size_t)cap); uint64_t tnext;
  int tn = brotli_oracle_probe_chain2(state, from, g_total_bits, tp, ti, td, cap, &tnext);
  { uint64_t lits = 0, dists = 0; int k = 0;
    if (getenv("WM_NOSWITCH")) { bl[0] = bl[2] = 1u << 30; bl[1] = (uint32_t)atoi(getenv("WM_NOSWITCH")); }   /* (as if block switches did not end a round: the parse behind one is not the stream's, its statistics are) */
    for (; k < tn; k++) { if ((uint32_t)k >= bl[1]) break; lits += ti[k]; dists += td[k]; if (lits > bl[0] || dists > bl[2]) break; }
    if (k < tn) tnext = tp[k];
    tn = k; }
  t_true += (uint64_t)tn; t_inv++;
  int done = 0;
  while (done < tn) {
    if (ti[done] > g_nmax) { t_byhand++; done++; continue; }
    const uint64_t entry = tp[done];
    /* this round: up to the next long run of the true chain (the engine hands it to the run regions) or the invocation's end */
    int upto = done; while (upto < tn && ti[upto] <= g_nmax) upto++;
    const uint64_t end = upto < tn ? tp[upto] : tnext;
    const int nseeds = (int)((end - entry + g_delta - 1) / g_delta);
    Chain* c = calloc((size_t)nseeds, sizeof *c);
    int32_t* mark_chain = malloc((size_t)(end - entry + 1) * 4); int32_t* mark_idx = malloc((size_t)(end - entry + 1) * 4);
    memset(mark_chain, 0xff, (size_t)(end - entry + 1) * 4);
    Lane* L = calloc((size_t)g_lanes, sizeof *L);
    int next_seed = 0; uint32_t tick = 0; const int waves = (g_lanes + 63) / 64;
    for (;;) {
      int any = 0;
      /* idle lanes take seeds in stream order */
      for (int l = 0; l < g_lanes; l++) if (L[l].state == 0 && next_seed < nseeds) {
        const int k = next_seed++;
        chain_more(state, &c[k], entry + (uint64_t)k * g_delta, end);
        c[k].link_chain = -1; L[l].chain = k; L[l].idx = 0; L[l].state = 1; t_seeds++;
      }
      for (int w = 0; w < waves; w++) {
        int wcmd = 0; uint32_t wlit = 0;
        for (int l = w * 64; l < (w + 1) * 64 && l < g_lanes; l++) {
          Lane* q = &L[l]; if (q->state == 0) continue;
          any = 1;
          Chain* ch = &c[q->chain];
          if (q->state == 1) {
            if (q->idx >= ch->n) chain_more(state, ch, ch->next, end);
            if (q->idx >= ch->n) { q->state = 0; ch->walked = q->idx; continue; }   /* ran out of the round's part */
            const uint64_t p = ch->pos[q->idx] - entry;
            if (mark_chain[p] >= 0) { ch->link_chain = mark_chain[p]; ch->link_idx = mark_idx[p]; ch->walked = q->idx; q->state = 0; continue; }
            if (ch->ins[q->idx] > g_nmax) { ch->walked = q->idx; ch->n = q->idx; q->state = 0; continue; }   /* a long run: the chain ends in front of it */
            mark_chain[p] = q->chain; mark_idx[p] = q->idx; ch->tick[q->idx] = tick;
            wcmd = 1; t_lane_cmd++;
            q->rem = ch->ins[q->idx]; q->state = 2;
          }
          if (q->state == 2) {
            uint32_t d = q->rem < (uint32_t)g_K ? q->rem : (uint32_t)g_K;
            q->rem -= d; t_lane_lit += d; if (d > wlit) wlit = d;
            if (q->rem == 0) { q->idx++; q->state = 1; }
          }
        }
        if (wcmd) { t_wave_cmd++; t_instr += g_A; }
        t_wave_lit += wlit; t_instr += g_B * wlit + (wcmd || wlit ? 10 : 0);
      }
      if (!any) break;
      tick++;
    }
    t_ticks += tick;
    /* stitch */
    int k = 0, idx = 0, got = 0, ok = 1; uint32_t crit = 0;
    for (;;) {
      for (int i = idx; i < c[k].walked; i++) {
        if (done + got >= upto) break;
        if (c[k].pos[i] != tp[done + got]) { ok = 0; break; }
        if (c[k].tick[i] > crit) crit = c[k].tick[i];
        got++;
      }
      if (!ok || done + got >= upto || c[k].link_chain < 0) break;
      idx = c[k].link_idx; k = c[k].link_chain;
    }
    if (!ok) { fprintf(stderr, "MISMATCH at round from bit %llu\n", (unsigned long long)entry); exit(1); }
    if (got == 0) { fprintf(stderr, "no progress at bit %llu\n", (unsigned long long)entry); exit(1); }
    if (done + got < upto && getenv("WM_VERBOSE")) printf("  round from %llu: got %d of %d (chain %d walked %d n %d)\n", (unsigned long long)entry, got, upto - done, k, c[k].walked, c[k].n);
    if (getenv("WM_VERBOSE")) printf("  round: entry %llu end %llu seeds %d (taken %d) true %d got %d ticks %u crit %u\n", (unsigned long long)entry, (unsigned long long)end, nseeds, next_seed, upto - done, got, tick, crit);
    t_ticks_crit += crit; t_rounds++;
    done += got;
    for (int q = 0; q < nseeds; q++) { free(c[q].pos); free(c[q].ins); free(c[q].dist); free(c[q].tick); }
    free(c); free(mark_chain); free(mark_idx); free(L);
  }
  free(tp); free(ti); free(td);
}
void oracle_stats_metablock(uint64_t first_bit, const void* state) { simulate(state, first_bit); }
void oracle_stats_switch(int category, uint64_t bit, uint64_t resume_bit, const void* state) { (void)category; (void)bit; if (!getenv("WM_NOSWITCH")) simulate(state, resume_bit); }
void oracle_stats_cmd(uint64_t a, uint64_t b, uint64_t c, int32_t d, int32_t e, uint32_t f, int32_t g, uint64_t h) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: walker_model <file.br> [delta_bits] [lanes] [K] [nmax]\n"); return 2; }
  if (argc > 2) g_delta = strtoull(argv[2], 0, 10);
  if (argc > 3) g_lanes = atoi(argv[3]);
  if (argc > 4) g_K = atoi(argv[4]);
  if (argc > 5) g_nmax = (uint32_t)atoi(argv[5]);
  FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* in = malloc((size_t)n + 8); if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2; fclose(f);
  g_total_bits = (uint64_t)n * 8;
  size_t cap = 256u << 20; uint8_t* out = malloc(cap);
  OracleInfo info; memset(&info, 0, sizeof info);
  brotli_oracle_decode(in, (size_t)n, out, cap, 1, &info);
  const double cmds = (double)(t_true ? t_true : 1);
  printf("%s: %llu commands, %.1f literals each; seeds every %llu bits, %d lanes, K %d: %llu invocations, %llu rounds, %llu seeds, by hand %llu\n", argv[1],
         (unsigned long long)info.num_commands, (double)info.num_literals / (double)info.num_commands, (unsigned long long)g_delta, g_lanes, g_K,
         (unsigned long long)t_inv, (unsigned long long)t_rounds, (unsigned long long)t_seeds, (unsigned long long)t_byhand);
  printf("  walker command steps %.2f and literals %.1f per true command; ticks %llu (critical %llu); wave command steps %llu, literal sub-steps %llu; %.1f wave instructions per true command\n",
         (double)t_lane_cmd / cmds, (double)t_lane_lit / cmds, (unsigned long long)t_ticks, (unsigned long long)t_ticks_crit, (unsigned long long)t_wave_cmd, (unsigned long long)t_wave_lit, t_instr / cmds);
  return 0;
}
