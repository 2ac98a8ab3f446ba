/* oracle/av1o_syms.h -- symbol sink used by both the rate estimator and the tile writer (test infrastructure). */
#ifndef ORACLE_AV1O_SYMS_H
#define ORACLE_AV1O_SYMS_H
#include "av1o_int.h"
typedef struct SymSink {
  void (*sym)(void *u, int cdf_off, int s, int nsyms);
  void (*lit)(void *u, uint32_t v, int nbits);
  void *u;
} SymSink;
void av1o_code_coeffs(const Av1oFrame *f, const int32_t *qc, int eob, int plane, int txs, int txtype,
                      int skip_ctx, int dc_ctx, int tx_cdf_off, int tx_sym, int tx_nsyms,
                      const SymSink *k, int *cul_level, int *dc_cat);
uint32_t av1o_coef_rate_full(const Av1oFrame *f, const int32_t *qc, int eob, int plane, int txs, int txtype,
                             int skip_ctx, int dc_ctx, int tx_cdf_off, int tx_sym, int tx_nsyms, int *cul, int *dccat);
int av1o_intra_tx_cdf(const Av1oFrame *f, int txs, int ymode, int *nsyms, int *set_out);
#endif
