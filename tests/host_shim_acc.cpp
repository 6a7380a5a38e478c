// Host shim of csrc/ccal_acc.hpp for tests/test_acc_host.py (CPU only; the product never runs this on the host).
#include "../cameracalibrations_amd/csrc/ccal_acc.hpp"
#include "../cameracalibrations_amd/csrc/ccal_math.hpp"

using namespace ccal;

extern "C" {

void shim_split(double x, int *j, long long *lo, long long *hi, int *flag) {
    int64_t l, h;
    acc_split(x, *j, l, h, *flag);
    *lo = l;
    *hi = h;
}

// adds xs[0..n) into R replicas (element i goes to replica i % R) exactly as the kernels do, then decodes.
// flags_out[0] = NaN contributions, [1] = out-of-range contributions.
// the 9x9 solve of the LM advance (csrc/ccal_lm.hip: lm_advance_wave), with the kernel's lanes as a loop:
// returns 1 if every pivot was positive (the positive-definiteness test), x[0..9) = S^-1 g.
int shim_gj_solve9(const double *S /* row-major 9x9 */, const double *g, double *x) {
    double a[9][9], b[9], diag[9];
    for (int i = 0; i < 9; i++) {
        for (int k = 0; k < 9; k++) a[i][k] = S[i * 9 + k];
        b[i] = g[i];
        diag[i] = 1.0;
    }
    int ok = 1;
    for (int j = 0; j < 9; j++) {
        const double d = a[j][j];
        ok = ok && (d > 0.0) && (d < 1.0e300);
        const double s = (1.0 / d) * (1.0 + 0x1p-20);   // any estimate of 1 / d will do: ~20 good bits here (v_rcp_f64 on the device)
        double prow[9];
        for (int k = 0; k < 9; k++) prow[k] = a[j][k];
        const double pb = b[j];
        for (int i = 0; i < 9; i++) gj_row_step<9>(j, i == j, d, s, prow, pb, a[i], b[i], diag[i]);
    }
    for (int i = 0; i < 9; i++) x[i] = b[i] / diag[i];
    return ok;
}

double shim_sum(const double *xs, int n, int R, int *flags_out) {
    static int64_t acc[ACC_MAX_REPLICAS][ACC_LIMBS];
    for (int r = 0; r < R; r++)
        for (int j = 0; j < ACC_LIMBS; j++) acc[r][j] = 0;
    flags_out[0] = flags_out[1] = 0;
    for (int i = 0; i < n; i++) {
        int j, flag;
        int64_t lo, hi;
        acc_split(xs[i], j, lo, hi, flag);
        if (flag) { flags_out[flag - 1]++; continue; }
        acc[i % R][j] += lo;
        if (hi) acc[i % R][j + 1] += hi;
    }
    // as the kernels read them back: every replica's word split into its two exactly representable halves, the halves added
    // as doubles and brought to canonical form (csrc/ccal_lm.hip acc_read_stage, csrc/ccal_init.hip foc_acc_read)
    AccPart part[ACC_LIMBS];
    for (int j = 0; j < ACC_LIMBS; j++) {
        part[j].hi = part[j].lo = 0.0;
        for (int r = 0; r < R; r++) acc_part_add(part[j], acc[r][j]);
        acc_part_norm(part[j]);
    }
    return acc_decode(part);
}
}

// the two rows of [Je | Ji | e] of one corner (csrc/ccal_math.hpp): the kernels' factored form (point_rows_K) and the
// plain form with the 3x9 Rodrigues Jacobian of SURVEY.md A.2 / A.3 (point_rows), for tests/test_acc_host.py
extern "C" void shim_point_rows(const double *M, double u, double v, const double *rvec, const double *tvec, const double *intr,
                                double aspect, double *rows_factored /* [2][16] */, double *rows_plain /* [2][16] */) {
    double R[9], dR[27], R2[9], K[9];
    rodrigues(rvec, R, dR);
    rodrigues_RK(rvec, R2, K);
    point_rows(M, u, v, R, dR, tvec, intr, aspect, rows_plain, rows_plain + 16);
    point_rows_K(M, u, v, R2, K, tvec, intr, aspect, rows_factored, rows_factored + 16, 1.0);
}
