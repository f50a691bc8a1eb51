/*
 * relmc_oracle.h — TEST INFRASTRUCTURE ONLY (see relmc_oracle.c).  CPU restatement of the
 * reference hot path; shares only the plain-data structs of include/relmc.h.
 */
#ifndef RELMC_ORACLE_H
#define RELMC_ORACLE_H
#include "../include/relmc.h"
#ifdef __cplusplus
extern "C" {
#endif
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_thresholds(const relmc_case_desc* c, uint32_t* thr);
int32_t orc_mc_sampling(const relmc_case_desc* c, uint64_t seed, uint64_t first_index, int64_t n,
                        uint8_t* eqstatus);
int32_t orc_mc_simulation(const relmc_case_desc* c, const uint8_t* states, int64_t n,
                          const relmc_solver_opts* opts, double* dns, double* nodal,
                          int32_t* status, int32_t* iters, int32_t* relaxed, int32_t nthreads);
int32_t orc_nsq_accumulate(const relmc_case_desc* c, uint64_t seed, uint64_t first_index, int64_t n,
                           const relmc_solver_opts* opts, int32_t nthreads, int32_t use_memo,
                           relmc_acc* acc_out);
int32_t orc_nsq_database(const relmc_case_desc* c, uint64_t seed, double beta_limit, int64_t max_iterations, int64_t samples_per_batch,
                         const relmc_solver_opts* opts, int32_t nthreads, int64_t max_rows,
                         uint8_t* db_states, int64_t* db_count, double* db_dns, int32_t* db_flag, double* db_nodal,
                         int32_t* db_status, int32_t* db_iters, int32_t* db_relaxed, int64_t* rows_out,
                         int64_t hist_cap, double* beta_hist, double* edns_hist, double* lole_hist, double* plc_hist,
                         int64_t* checkpoints_out, int64_t* iterations_out, relmc_acc* acc_out);
void orc_nsq_indices(const relmc_acc* a, int32_t nb, int32_t ncomp, double hours, relmc_indices* out);
int32_t orc_seq_mcsimulation(const relmc_case_desc* c, const uint8_t* states, const double* load_scale, int64_t n,
                             const relmc_solver_opts* opts, double* dns, double* nodal,
                             int32_t* status, int32_t* iters, int32_t* relaxed, int32_t nthreads);
int32_t orc_seq_mcsampling(int32_t ncomp, const double* mttf, const double* mttr, int32_t hpy, uint64_t seed,
                           uint64_t first_year, int32_t num_years, uint8_t* out);
int32_t orc_seq_years(const relmc_case_desc* c, const double* mttf, const double* mttr, int32_t hpy, const double* load_factors,
                      uint64_t seed, uint64_t first_year, int32_t n_years, const relmc_solver_opts* opts, double threshold,
                      int32_t nthreads, double* years_out, relmc_acc* acc);
int32_t orc_hl1_nsq(int32_t ngen, const double* cap, const double* for_rate, int32_t nhours, const double* load,
                    uint64_t seed, uint64_t first_index, int64_t n, double* iter_lole, double* iter_eue);
int32_t orc_max_threads(void);
#ifdef __cplusplus
}
#endif
#endif
