// ndt_pass.h — internal: kernel argument block of one derivative pass.
#ifndef LSA_NDT_PASS_H_
#define LSA_NDT_PASS_H_

struct lsa_ndt;

struct PassParams {
  float T[12];       // rigid transform, row-major 3x4, f32 as the reference builds it
  double ja[8][3];   // j_ang_a .. j_ang_h   (NormalDistributionsTransform.cpp:551-581)
  double ha[15][3];  // h_ang_a2,a3,b2,b3,c2,c3,d1,d2,d3,e1,e2,e3,f1,f2,f3 (:584-642)
  double d1, d2;     // gauss_d1_, gauss_d2_
  double r2max;      // largest double x with sqrt(x) < (double)res: `sqrt(d2) < radius` == `d2 <= r2max`, no sqrt per probe
  float res;         // resolution_ = search radius = leaf
  float inv_res;     // 1 / res, used only when res_pow2 (then x / res == x * inv_res bit for bit)
  int res_pow2;
  float r2f_pre;     // dense-table probe: f32 squared-distance pre-filter threshold (r^2 + rounding margin)
  int min_b[3];      // real_min_b*
  int dim[3];        // real_max_b* - real_min_b* + 1
};

void ndt_make_pass_params(const lsa_ndt *h, const float T[16], const double p[6], PassParams *P);
unsigned ndt_pass_grid(const lsa_ndt *h, size_t n, int kind);
int ndt_enqueue_pass(lsa_ndt *h, const PassParams &P, int kind, bool reduce = true);
bool ndt_pass_host_reduce(const lsa_ndt *h, int kind);
int ndt_publish_out(lsa_ndt *h, unsigned long long *key_out);
int ndt_enqueue_pass_host(lsa_ndt *h, const PassParams &P, int kind, unsigned *grid_out, unsigned long long *key_out);
struct PassMail;
void ndt_mail_post(PassMail *mail, const PassParams *P, unsigned long long seq, bool cancel, const void *d_src, size_t n, unsigned vgrid, bool kept = false);
int ndt_launch_armed(lsa_ndt *h, const PassParams &P, unsigned long long seq, unsigned grid, double patience_us);
int ndt_reserve_armed(lsa_ndt *h);
unsigned long long ndt_armed_row_key(unsigned long long seq);
double *ndt_armed_rows(lsa_ndt *h, unsigned long long seq);
bool ndt_fetch_row_host(const double *row, unsigned long long key, double limit_ms, unsigned long long out32[32],
                        const volatile unsigned long long *abort_word = nullptr, unsigned long long abort_value = 0);
bool ndt_reduce_partials_host(const double *partials, unsigned nb, unsigned long long key, double limit_ms, double out32[32],
                              const volatile unsigned long long *abort_word = nullptr, unsigned long long abort_value = 0);
// Host-stall detector.  The row spin loop reads the time-stamp counter every iteration and keeps, per calling thread, the longest
// interval between two consecutive looks; the armed paths add the interval from posting a pass to the first look.  An interval of a
// millisecond means the OS held the thread up (a look takes well under a microsecond): what tells a late or redone armed pass that is
// the box's doing from one that is the library's.  Returns the maximum since the last reset, in microseconds.
double lsa_host_gap_take_us(bool reset);
// the longest wait for a single row since the last reset (first look -> accepted or given up), microseconds: the quantity the per-row
// limits bound.  A pass's rows_wait_max_us is the sum over its rows as they trickle in and may exceed the limit when the DEVICE is slow.
double lsa_row_wait_take_us(bool reset);
void lsa_host_gap_note_us(double us);

#endif
