/* Exhaustive check behind navsim.hip's div_const(): for the operand domains the step kernel feeds it, one multiply by the
 * rounded reciprocal plus one fma correction equals the correctly rounded IEEE quotient.
 *   gcc -O2 -ffp-contract=off -o /tmp/divc tools/verify/div_by_const_exact.c -lm && /tmp/divc      (expects "bad=0" twice) */
#include <math.h>
#include <stdio.h>
static int ok(double q, double s) {
    const double inv = 1.0 / s, r0 = q * inv, e = fma(-s, r0, q), r1 = fma(e, inv, r0);
    return r1 == q / s;
}
int main(void) {
    long bad = 0;
    /* py_round(x, n): integer-valued q with |q| < 2^26, s = 10 and 100 */
    for (long q = -(1L << 26); q <= (1L << 26); ++q) bad += !ok((double)q, 10.0) + !ok((double)q, 100.0);
    printf("py_round quotients: bad=%ld\n", bad);
    bad = 0;
    /* observation tail: yaw (integer degrees) / 360, rel_theta (k/100) / 360, diff_angle (k/100) / 180 */
    for (int y = -720; y <= 720; ++y) bad += !ok((double)y, 360.0);
    for (long k = -72000; k <= 72000; ++k) bad += !ok((double)k / 100.0, 360.0) + !ok((double)k / 100.0, 180.0);
    printf("observation quotients: bad=%ld\n", bad);
    return 0;
}
