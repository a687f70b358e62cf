/* Prints sizeof / offsetof of the two structs that cross the C-ABI of libgdca.so (include/gdca.h), one
 * "struct field offset size" line per field.  tests/test_cabi_cpu.py compiles this with gcc and compares the
 * layout with the ctypes mirror (gaussdca.jl_amd/_lib.py) and with the Julia mirror (julia/GaussDCAHip.jl). */
#include <stddef.h>
#include <stdio.h>

#include "gdca.h"

#define F(S, f) printf(#S " " #f " %zu %zu\n", offsetof(S, f), sizeof(((S *)0)->f))

int main(void)
{
    printf("gdca_params . 0 %zu\n", sizeof(gdca_params));
    F(gdca_params, pseudocount);
    F(gdca_params, theta);
    F(gdca_params, score);
    F(gdca_params, apc);
    printf("gdca_stats . 0 %zu\n", sizeof(gdca_stats));
    F(gdca_stats, theta);
    F(gdca_stats, Meff);
    F(gdca_stats, pair_identity_sum);
    F(gdca_stats, thresh);
    F(gdca_stats, info);
    F(gdca_stats, N);
    F(gdca_stats, M);
    F(gdca_stats, q);
    F(gdca_stats, n);
    F(gdca_stats, n_pad);
    F(gdca_stats, update_launches);
    F(gdca_stats, inverse_batch);
    F(gdca_stats, refined);
    F(gdca_stats, ms_total);
    F(gdca_stats, ms_theta);
    F(gdca_stats, ms_weights);
    F(gdca_stats, ms_covariance);
    F(gdca_stats, ms_inverse);
    F(gdca_stats, ms_inverse_update);
    F(gdca_stats, ms_score);
    F(gdca_stats, inverse_flops);
    F(gdca_stats, update_flops);
    F(gdca_stats, sweep_ghz);
    F(gdca_stats, inverse_norm1);
    F(gdca_stats, matrix_norm1);
    F(gdca_stats, cond_bound);
    F(gdca_stats, ms_fn);
    F(gdca_stats, ms_pair_tally);
    F(gdca_stats, sweep_retries);
    F(gdca_stats, reserved0);
    printf("status GDCA_OK %d 0\nstatus GDCA_EINVAL %d 0\nstatus GDCA_ENOTPD %d 0\nstatus GDCA_EHIP %d 0\n"
           "status GDCA_ENOMEM %d 0\nstatus GDCA_ENOCONV %d 0\n",
           GDCA_OK, GDCA_EINVAL, GDCA_ENOTPD, GDCA_EHIP, GDCA_ENOMEM, GDCA_ENOCONV);
    return 0;
}
