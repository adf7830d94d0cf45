/* c99_include.c -- include/rawdtw.h must be consumable from plain C99 (test infrastructure: compiled with
 * gcc -std=c99 -pedantic-errors -fsyntax-only by tests/test_abi_shim.py). */
#include "rawdtw.h"

int use_the_abi(void)
{
    rawdtw_job_t j = {0, 0, 1, 1, RAWDTW_FULL, 0, 0};
    rawdtw_align_opt_t o = {1, 1, 0.10f, 0.4f, 20.0f, 1};
    rawdtw_anchor_t a[2] = {{5, 5}, {1, 1}};
    rawdtw_job_t out[1];
    (void)j;
    return rawdtw_abi_version() == RAWDTW_ABI_VERSION && rawdtw_chain_job_count(&o, 2) == 1 &&
           rawdtw_chain_build_jobs(&o, a, 2, 0, 0, 0, out) == RAWDTW_OK;
}
