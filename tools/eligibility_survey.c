/* Which metablocks can which command engine take?  Built against the oracle with -DORACLE_STATS (tools/eligibility_survey.py);
 * reads "<len>\n<bytes>" records of Brotli streams from stdin, decodes each, and counts per stream the metablocks (by their
 * uncompressed length) whose literals do not depend on context -- one literal tree per block type -- and, of those, the ones
 * whose distance contexts use more than one prefix code (the scan engine's: DESIGN 2c). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int32_t result, error_code; uint64_t decoded_size, consumed, num_commands, num_metablocks; uint32_t max_literal_trees, r; } Info;
extern int brotli_oracle_decode(const uint8_t*, size_t, uint8_t*, size_t, uint32_t, void*);
extern int brotli_oracle_metablock_shape(const void* state, uint32_t out[4]);
static uint64_t tot, ctxfree, ctxfree_multi, multi, sw[3], ncmd, ndict;
void oracle_stats_metablock(uint64_t first_bit, const void* state) {
  uint32_t o[4]; (void)first_bit;
  brotli_oracle_metablock_shape(state, o);   /* mlen, literal trees per block type > 1 ?, distance contexts differ ?, 0 */
  tot += o[0];
  if (!o[1]) { ctxfree += o[0]; if (o[2]) ctxfree_multi += o[0]; }
  if (o[2]) multi += o[0];
}
void oracle_stats_switch(int c, uint64_t b, uint64_t r, const void* s) { (void)b; (void)r; (void)s; if (c >= 0 && c < 3) sw[c]++; }
void oracle_stats_cmd(uint64_t a, uint64_t b, uint64_t c, int32_t d, int32_t e, uint32_t f, int32_t g, uint64_t h) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)h; ncmd++; if (g > 0 && (uint64_t)g > (h < ((1ull << 22) - 16) ? h : ((1ull << 22) - 16))) ndict++; }
int main(void) {
  char name[256]; size_t n;
  while (scanf("%255s %zu", name, &n) == 2) {
    getchar();
    uint8_t* d = malloc(n); if (fread(d, 1, n, stdin) != n) return 1;
    uint8_t* out = malloc(64u << 20);
    uint8_t info[256];
    tot = ctxfree = ctxfree_multi = multi = 0; sw[0] = sw[1] = sw[2] = ncmd = ndict = 0;
    brotli_oracle_decode(d, n, out, 64u << 20, 1, info);
    printf("%-44s bytes %9llu: literals context-free %5.1f %% (of those, per-context distance codes %5.1f %%); per-context distance codes %5.1f %%; %llu commands, block switches L/C/D %llu/%llu/%llu: %.0f commands between two switches; words of the static dictionary: one command in %.0f\n", name,
           (unsigned long long)tot, tot ? 100.0 * ctxfree / tot : 0.0, ctxfree ? 100.0 * ctxfree_multi / ctxfree : 0.0, tot ? 100.0 * multi / tot : 0.0, (unsigned long long)ncmd, (unsigned long long)sw[0], (unsigned long long)sw[1], (unsigned long long)sw[2], (double)ncmd / (double)(sw[0] + sw[1] + sw[2] + 1), ndict ? (double)ncmd / (double)ndict : 0.0);
    free(d); free(out);
  }
  return 0;
}
