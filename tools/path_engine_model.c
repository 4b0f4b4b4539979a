/* path_engine_model -- CPU model of pass 1 of the path engine (csrc/brotli_path_engine.h), checked against the oracle.
 *
 * Analysis / test tool, not product code: it includes the oracle's source (built with -DORACLE_STATS) to use its tables,
 * re-plays what the engine's first pass does region by region -- literal code lengths at every bit (J1), the literal
 * path from the region's entry with its ranks, the command records of every state (path position, "a distance code
 * starts here") plus the closure over the states those records lead to (off the path, or "a command head starts here"),
 * and the walk along next[] -- and compares every command the walk finds (position, insert length, copy length,
 * distance symbol) with the command the oracle decodes there.  Prints the work counts the design is sized by.
 *
 * usage: path_engine_model <file.br> [region_bits=32768] [hop_cap=16]
 */
#define ORACLE_STATS
#include "../oracle/brotli_oracle.c"
#include <stdio.h>

static uint32_t RBL = 32768, HOPCAP = 16, RANKCAP = 6656, WCAP = 4096;
static int g_byhand_eval; /* the walker's own evaluation: no hop cap, no closure state (returns ST_NONE with the state in g_bh_pos / g_bh_v when the next state is not a path state) */
static uint32_t g_bh_pos; static int g_bh_v;
enum { ST_END = 0xFFFF, ST_BYHAND = 0xFFFE, ST_NONE = 0xFFFD };
typedef struct { uint64_t p; uint32_t ins, copy, dsym; int implicit; } Cmd;

/* region */
static const DS* g_s;
static uint64_t g_lb;            /* absolute bit of local position 0 */
static uint32_t g_L, g_Lp, g_Rn; /* parse limit, path limit, ranks */
static uint8_t* g_J1;            /* len | 0x80 if on path */
static uint16_t* g_por;          /* position of rank */
static uint16_t* g_rank_at;      /* model shortcut for rank(y): the GPU computes it from chunk masks */
static uint16_t* g_next;         /* RANKCAP + WCAP states */
static uint16_t* g_wstate; static uint32_t g_wn;
static Cmd* g_pred; static uint32_t g_npred, g_ipred; /* commands predicted by the current region's walk */
static int g_valid;              /* a region is alive */
static uint64_t st_regions, st_cmds, st_evals_base, st_evals_work, st_rounds, st_byhand_states, st_byhand_hit, st_end_noprogress, st_hops_hist[40], st_mismatch, st_checked, st_unpredicted;
static uint64_t st_wmax, st_rankmax, st_trunc_rank, st_bh_hop, st_bh_wcap, st_w_imp, st_w_off, st_w_dup, st_end_lim, st_end_y, st_end_rank;
static uint8_t* g_seen;

static inline uint32_t peek_at(uint64_t abs_bit) { BR br = g_s->br; br.pos = abs_bit; return (uint32_t)br_peek(&br); }
static inline uint64_t peek64_at(uint64_t abs_bit) { BR br = g_s->br; br.pos = abs_bit; return br_peek(&br); }
static uint32_t sym_at(const HC* table, uint64_t abs_bit, uint32_t* len) {
  BR br = g_s->br; br.pos = abs_bit; uint32_t v; read_symbol(&br, table, &v); *len = (uint32_t)(br.pos - abs_bit); return v;
}
static const HC* lit_table(void) { return g_s->lit_codes + g_s->lit_htrees[g_s->context_map[g_s->block_type_rb[1] << 6]]; }
static const HC* cmd_table(void) { return g_s->cmd_codes + g_s->cmd_htrees[g_s->block_type_rb[3]]; }
static const HC* dist_table(void) { return g_s->dist_codes + g_s->dist_htrees[g_s->dist_context_map[g_s->block_type_rb[5] << 2]]; }

/* distance code at local q: symbol and total bits */
static uint32_t dist_parse(uint32_t q, uint32_t* bits) {
  uint32_t len; uint32_t code = sym_at(dist_table(), g_lb + q, &len);
  uint32_t nb = 0;
  if (code >= 16) { int32_t dv = (int32_t)code - (int32_t)g_s->num_direct; if (dv >= 0) { dv >>= g_s->postfix_bits; nb = ((uint32_t)dv >> 1) + 1; } }
  *bits = len + nb; return code;
}
/* one record: state (pos, v) -> next state id; fills *c with the command whose head it parses */
static uint16_t eval_state(uint32_t pos, int v, Cmd* c, uint32_t* dsym_prev) {
  uint32_t q = pos, p = q;
  if (q + 128 > g_L) return ST_END;
  if (v == 0) { uint32_t db; uint32_t code = dist_parse(q, &db); if (dsym_prev) *dsym_prev = code; p = q + db; }
  uint32_t len; uint32_t cmd = sym_at(cmd_table(), g_lb + p, &len);
  CmdLut lut = g_cmd_lut[cmd];
  uint64_t w = peek64_at(g_lb + p) >> len;
  uint32_t ins = lut.ins_off + (uint32_t)(w & ((1ull << lut.ins_extra) - 1)); w >>= lut.ins_extra;
  uint32_t copy = lut.copy_off + (uint32_t)(w & ((1ull << lut.copy_extra) - 1));
  uint32_t hbits = len + lut.ins_extra + lut.copy_extra;
  int imp = cmd < 128;
  if (c) { c->p = g_lb + p; c->ins = ins; c->copy = copy; c->implicit = imp; c->dsym = 0xFFFF; }
  uint32_t y = p + hbits, n = ins, hops = 0;
  while (n > 0 && y < g_Lp && !(g_J1[y] & 0x80) && (g_byhand_eval || hops < HOPCAP)) { y += g_J1[y] & 15; n--; hops++; }
  st_hops_hist[hops < 39 ? hops : 39]++;
  if (y >= g_Lp) return ST_END;
  uint32_t q2;
  if (n > 0) {
    if (!(g_J1[y] & 0x80)) { st_bh_hop++; return ST_BYHAND; }
    uint32_t r = g_rank_at[y] + n;
    if (r >= g_Rn) return ST_END;
    q2 = g_por[r];
  } else q2 = y;
  if (!imp && (g_J1[q2] & 0x80)) return g_rank_at[q2];
  if (g_byhand_eval) { g_bh_pos = q2; g_bh_v = imp; return ST_NONE; }
  if (g_wn >= WCAP) { st_bh_wcap++; return ST_BYHAND; }
  if (imp) st_w_imp++; else st_w_off++;
  { uint32_t key = q2 * 2 + (imp ? 1 : 0); if (g_seen[key >> 3] >> (key & 7) & 1) st_w_dup++; g_seen[key >> 3] |= (uint8_t)(1u << (key & 7)); }
  g_wstate[g_wn] = (uint16_t)(q2 | (imp ? 0x8000u : 0u));
  return (uint16_t)(RANKCAP + g_wn++);
}
static void build_region(const DS* s, uint64_t entry_bit) {
  g_s = s; st_regions++;
  g_lb = entry_bit & ~31ull;
  uint64_t avail = s->br.total_bits - g_lb;
  g_L = avail < RBL ? (uint32_t)avail : RBL;
  const uint32_t le = (uint32_t)(entry_bit - g_lb);
  const HC* lt = lit_table();
  for (uint32_t l = 0; l < RBL; l++) { uint32_t len; sym_at(lt, g_lb + l, &len); g_J1[l] = (uint8_t)len; }
  /* the literal path from the entry */
  g_Rn = 0; g_Lp = g_L > 16 ? g_L - 16 : 0;
  for (uint32_t y = le; y < g_Lp; y += g_J1[y] & 15) {
    if (g_Rn >= RANKCAP) { g_Lp = y; st_trunc_rank++; break; }
    g_rank_at[y] = (uint16_t)g_Rn; g_por[g_Rn++] = (uint16_t)y; g_J1[y] |= 0x80;
  }
  if (g_Rn > st_rankmax) st_rankmax = g_Rn;
  /* records: the entry state first (a command head starts at the entry), then the base states, then the closure */
  g_wn = 0; g_wstate[g_wn++] = (uint16_t)(le | 0x8000u); memset(g_seen, 0, RBL / 4 + 16);
  for (uint32_t r = 0; r < g_Rn; r++) { g_next[r] = eval_state(g_por[r], 0, NULL, NULL); st_evals_base++; }
  uint32_t done = 0; int rounds = 0;
  while (done < g_wn) {
    uint32_t end = g_wn; rounds++;
    for (uint32_t k = done; k < end; k++) { g_next[RANKCAP + k] = eval_state(g_wstate[k] & 0x7FFF, (g_wstate[k] >> 15) ? 1 : 0, NULL, NULL); st_evals_work++; }
    done = end;
  }
  st_rounds += (uint64_t)rounds; if (g_wn > st_wmax) st_wmax = g_wn;
  if (getenv("PEM_REGIONS")) fprintf(stderr, "region %llu at bit %llu: ranks %u closure %u rounds %d\n", (unsigned long long)st_regions, (unsigned long long)entry_bit, g_Rn, g_wn, rounds);
  /* the walk: commands of the true chain as this region sees them.  A state whose record says BYHAND (hop cap, closure
   * full) is evaluated by the walker itself, and so are the states behind it until the chain is back on a path state. */
  g_npred = 0; g_ipred = 0;
  uint32_t pos = le; int v = 1; uint16_t id = (uint16_t)RANKCAP; int have_id = 1;
  for (;;) {
    Cmd c; uint32_t dprev = 0xFFFF;
    uint32_t wsave = g_wn;
    g_byhand_eval = 1; uint16_t mine = eval_state(pos, v, &c, &dprev); g_byhand_eval = 0; g_wn = wsave;
    uint16_t stored = have_id ? g_next[id] : ST_BYHAND;
    if (v == 0 && g_npred) g_pred[g_npred - 1].dsym = dprev;  /* the distance of the command before */
    if (stored == ST_END || mine == ST_END) break;
    g_pred[g_npred++] = c;
    if (stored == ST_BYHAND) {
      st_byhand_hit++;
      if (mine == ST_NONE) { pos = g_bh_pos; v = g_bh_v; have_id = 0; } else { id = mine; have_id = 1; pos = g_por[id]; v = 0; }
      continue;
    }
    id = stored; have_id = 1;
    if (id >= RANKCAP) { pos = g_wstate[id - RANKCAP] & 0x7FFF; v = g_wstate[id - RANKCAP] >> 15; } else { pos = g_por[id]; v = 0; }
  }
  /* the last command's distance is in the record of the state the walk stopped at: the model drops that command (the
   * engine drops it too unless the distance can be parsed) */
  if (g_npred) g_npred--;
  if (g_npred == 0) st_end_noprogress++;
  g_valid = 1;
}
static const DS* g_cur; static int g_switched;
void oracle_stats_metablock(uint64_t first_bit, const void* state) { (void)first_bit; g_cur = (const DS*)state; g_valid = 0; }
void oracle_stats_switch(int category, uint64_t bit, uint64_t resume_bit, const void* state) { (void)category; (void)bit; (void)resume_bit; (void)state; g_valid = 0; g_switched = 1; }
void oracle_stats_cmd(uint64_t cmd_pos, uint64_t lit_pos, uint64_t end_pos, int32_t ins, int32_t copy, uint32_t dsym, int32_t dist, uint64_t P) {
  (void)lit_pos; (void)end_pos; (void)dist; (void)P;
  st_cmds++;
  if (g_switched) { g_switched = 0; g_valid = 0; st_unpredicted++; return; }  /* a block switch inside this command: the checked loop's */
  if (!g_valid || g_ipred >= g_npred) {
    /* the engine would be entered here -- but only for metablocks whose literals do not depend on context and whose four distance contexts share a tree */
    const DS* s = g_cur;
    uint32_t bt = s->block_type_rb[1]; int trivial = (s->trivial[bt >> 5] >> (bt & 31)) & 1;
    const uint8_t* dm = s->dist_context_map + (s->block_type_rb[5] << 2);
    if (!trivial || dm[0] != dm[1] || dm[0] != dm[2] || dm[0] != dm[3] || s->large_window) { g_valid = 0; st_unpredicted++; return; }
    build_region(s, cmd_pos);
    if (g_npred == 0) { g_valid = 0; st_unpredicted++; return; }
  }
  Cmd* c = &g_pred[g_ipred++];
  st_checked++;
  uint32_t ds = dsym == 0xFFFF ? 0xFFFF : (dsym & 0xFFF);
  if (c->p != cmd_pos || c->ins != (uint32_t)ins || c->copy != (uint32_t)copy || (c->dsym != ds)) {
    if (st_mismatch < 10) fprintf(stderr, "MISMATCH at command %llu: predicted p=%llu ins=%u copy=%u dsym=%u, oracle p=%llu ins=%d copy=%d dsym=%u\n", (unsigned long long)st_cmds,
                                  (unsigned long long)c->p, c->ins, c->copy, c->dsym, (unsigned long long)cmd_pos, ins, copy, ds);
    st_mismatch++; g_valid = 0;
  }
}
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: path_engine_model <file.br> [region_bits] [hop_cap]\n"); return 2; }
  if (argc > 2) RBL = (uint32_t)atoi(argv[2]);
  if (argc > 3) HOPCAP = (uint32_t)atoi(argv[3]);
  RANKCAP = RBL * 13 / 64; WCAP = getenv("PEM_WCAP") ? (uint32_t)atoi(getenv("PEM_WCAP")) : RBL / 4;
  FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* in = malloc((size_t)n + 8); if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2; fclose(f);
  g_J1 = malloc(RBL + 64); g_seen = malloc(RBL / 4 + 16); g_por = malloc(2 * (RANKCAP + 1)); g_rank_at = malloc(2 * (RBL + 64)); g_next = malloc(2 * (RANKCAP + WCAP)); g_wstate = malloc(2 * WCAP); g_pred = malloc(sizeof(Cmd) * (RANKCAP + WCAP + 1));
  size_t cap = 256u << 20; uint8_t* out = malloc(cap);
  OracleInfo info; memset(&info, 0, sizeof info);
  DS dummy; (void)dummy;
  brotli_oracle_decode(in, (size_t)n, out, cap, 1, &info);
  printf("stream: %ld -> %llu bytes, %llu commands; regions of %u bits, %u ranks, %u closure states, %u hops\n", n, (unsigned long long)info.decoded_size, (unsigned long long)info.num_commands, RBL, RANKCAP, WCAP, HOPCAP);
  printf("regions %llu (%.1f commands each), commands checked %llu, MISMATCHES %llu, commands outside the engine %llu, regions without progress %llu\n", (unsigned long long)st_regions,
         (double)st_checked / (double)(st_regions ? st_regions : 1), (unsigned long long)st_checked, (unsigned long long)st_mismatch, (unsigned long long)st_unpredicted, (unsigned long long)st_end_noprogress);
  printf("record evaluations per region: %.0f path states + %.0f closure states in %.1f rounds (most closure states %llu, most ranks %llu, regions cut by the rank cap %llu); states the walks took by hand %llu\n",
         (double)st_evals_base / (double)st_regions, (double)st_evals_work / (double)st_regions, (double)st_rounds / (double)st_regions, (unsigned long long)st_wmax, (unsigned long long)st_rankmax, (unsigned long long)st_trunc_rank, (unsigned long long)st_byhand_hit);
 printf("by-hand records: hop cap %llu, closure full %llu; closure states: implicit %llu, off the path %llu, of which duplicates %llu\n", (unsigned long long)st_bh_hop, (unsigned long long)st_bh_wcap, (unsigned long long)st_w_imp, (unsigned long long)st_w_off, (unsigned long long)st_w_dup);
  printf("hops before the path is met:"); { uint64_t tot = 0, acc = 0; for (int k = 0; k < 40; k++) tot += st_hops_hist[k]; for (int k = 0; k <= (int)HOPCAP && k < 40; k++) { acc += st_hops_hist[k]; printf(" <=%d %.1f%%", k, 100.0 * (double)acc / (double)tot); } printf("\n"); }
  return st_mismatch != 0;
}
