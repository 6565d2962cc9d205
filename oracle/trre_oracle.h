/*
 * trre_oracle.h — CPU oracle for the scan-mode hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (trre_amd/, libtrre_mi355x.so) never links,
 * imports or executes anything under oracle/.
 *
 * Plain-C restatement of the reference's algorithm for the scan path:
 *   pattern -> AST            (trre_nft.c:11-288 / trre_dft.c:13-288)
 *   AST -> NFT                (trre_nft.c:323-511 / trre_dft.c:322-523)
 *   NFT backtracking attempt  (trre_nft.c:593-657)
 *   lazy determinisation      (trre_dft.c:874-1010, 1135-1175)
 *   DFT attempt               (trre_dft.c:1110-1196)
 *   scan line loops           (trre_nft.c:775-790 / trre_dft.c:1272-1286)
 *
 * Parity pin: checked byte-for-byte against the compiled reference binaries
 * (oracle/_ref/trre, oracle/_ref/trre_dft) on the golden fixtures under
 * tests/golden/ and by the randomized differential script tests/fuzz_oracle.py.
 */
#ifndef TRRE_ORACLE_H
#define TRRE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TRRE_ORACLE_NFT = 0, TRRE_ORACLE_DFT = 1 };

typedef struct trre_oracle_prog trre_oracle_prog;

/* Compile a pattern.  Returns 0 on success; on failure returns a negative code
 * and writes the reference's stderr text (without trailing newline) to err. */
int trre_oracle_compile(const char *pattern, int engine, trre_oracle_prog **out,
                        char *err, size_t errcap);

/* Scan a whole '\n'-delimited buffer the way the reference's main() does for
 * a FILE.  *out is malloc'ed (caller frees with trre_oracle_release).
 * Returns 0, or a negative code (e.g. backtracking stack overflow, which makes
 * the reference exit(1)). */
int trre_oracle_scan(trre_oracle_prog *p, const uint8_t *in, size_t n,
                     uint8_t **out, size_t *m);

/* `trre -m PATTERN` (match mode, trre_nft.c:791-797 with 635-642): per record one attempt over the whole line,
 * accepted only at its end; prints the attempt's output and '\n', nothing for a line that does not match.
 * NFT engine only. */
int trre_oracle_match(trre_oracle_prog *p, const uint8_t *in, size_t n, uint8_t **out, size_t *m);

/* `trre -a` / `trre -ma` (generator mode, trre_nft.c:736-738 with 640-641, 647-648): after this call
 * trre_oracle_scan / trre_oracle_match print the output of EVERY accepting path, in the search's depth-first
 * priority order (scan mode: all outputs of the attempt at a position, then that position's raw byte).  NFT engine only. */
int trre_oracle_set_all(trre_oracle_prog *p, int all);

/* Line-sharded scan over `threads` host threads (input split at '\n'
 * boundaries, outputs concatenated in order).  Each thread compiles its own
 * program because the lazily grown DFT cache is not thread-safe. */
int trre_oracle_scan_mt(const char *pattern, int engine, int threads,
                        const uint8_t *in, size_t n, uint8_t **out, size_t *m);

void trre_oracle_release(uint8_t *buf);
void trre_oracle_free(trre_oracle_prog *p);

/* introspection used by tests */
int trre_oracle_nft_states(const trre_oracle_prog *p);
int trre_oracle_dft_states(const trre_oracle_prog *p);   /* explored so far */

#ifdef __cplusplus
}
#endif
#endif
