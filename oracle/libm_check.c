/* oracle/libm_check.c -- TEST INFRASTRUCTURE.  Proves that simplestereo_amd/csrc/glibc_math.hip.h (glibc's exp and powf
 * restated for the device: tables from tools/extract_glibc_tables.py, fused multiply-adds placed as the library's FMA build
 * has them) returns the SAME BITS as the libm this process runs on -- the libm the reference's _passive extension links
 * (reference _passive.cpp:47-50, 360-364: exp; headers/colorconversion.hpp:55-65: powf).
 *
 *   gcc -O2 -ffp-contract=off -I../simplestereo_amd/csrc libm_check.c -o libm_check -lm && ./libm_check [millions]
 *
 * exp: n random arguments of the range the support weights use ((-60, 0], dense near 0), of (-760, 0] (through the
 * rescaled branch and into the subnormals), tiny and huge ones and the special values.  powf(x, (float)(1 / 3.0)): EVERY float
 * of [0.008856, 1.3] -- the only call the Lab conversion makes beyond its 256-entry byte table.  Exit code 0 and a line
 * "libm_check ok ..." when no argument differs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "glibc_math.hip.h"

static uint64_t s_ = 88172645463325252ull;
static inline uint64_t rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
static inline double uni(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    const long n = (argc > 1 ? atol(argv[1]) : 20) * 1000000L;
    long bad_exp = 0, bad_powf = 0, n_powf = 0;
    for (long k = 0; k < n; ++k) {
        double x;
        switch (k % 4) {
        case 0: x = -60.0 * uni() * uni(); break;
        case 1: x = -760.0 * uni(); break;
        case 2: x = -500.0 - 260.0 * uni(); break;
        default: x = (k % 8 == 3) ? 1e-17 * uni() - 5e-18 : -1100.0 * uni() * uni() + 2.0 * uni(); break;
        }
        if (gm_asu64(exp(x)) != gm_asu64(glibc_exp(x))) {
            if (bad_exp < 5) printf("exp(%a): libm %a restatement %a\n", x, exp(x), glibc_exp(x));
            ++bad_exp;
        }
    }
    const double special[] = {0.0, -0.0, -INFINITY, INFINITY, -745.2, -746.0, -1023.9, -1024.0, -5000.0, 709.7, 710.0, 1e-300, -1e-300, 512.0, -512.0};
    for (unsigned k = 0; k < sizeof(special) / sizeof(special[0]); ++k)
        if (gm_asu64(exp(special[k])) != gm_asu64(glibc_exp(special[k]))) {
            printf("exp(%a): libm %a restatement %a\n", special[k], exp(special[k]), glibc_exp(special[k]));
            ++bad_exp;
        }
    const float third = 1 / 3.0;
    for (uint32_t u = gm_asu32(0.008856f); u <= gm_asu32(1.3f); ++u, ++n_powf) {
        const float x = gm_asf32(u);
        if (gm_asu32(powf(x, third)) != gm_asu32(glibc_powf_pos(x, third))) {
            if (bad_powf < 5) printf("powf(%a, 1/3): libm %a restatement %a\n", x, powf(x, third), glibc_powf_pos(x, third));
            ++bad_powf;
        }
    }
    /* the other exponent of the conversion (2.4, colorconversion.hpp:25-35) goes through a 256-entry table built by the host's
       powf itself; checked here all the same on the 256 arguments */
    for (int v = 0; v < 256; ++v) {
        float c = v / 255.0;
        if (c > 0.04045) {
            const float a = (c + 0.055) / 1.055;
            if (gm_asu32(powf(a, 2.4)) != gm_asu32(glibc_powf_pos(a, 2.4))) ++bad_powf;
        }
    }
    /* the proximity tables of the ASW path (reference _passive.cpp:360-364: exp(-sqrt(di^2 + dj^2) / gammaP)): since round 6 the
       library builds them with the restated exp on the host side -- every argument of every window up to 255 x 255, several gammaP */
    long n_prox = 0;
    {
        const double gps[] = {17.5, 3.0, 100.0, 0.5, 36.0};
        for (unsigned g = 0; g < sizeof(gps) / sizeof(gps[0]); ++g)
            for (int di = 0; di <= 127; ++di)
                for (int dj = di; dj <= 127; ++dj, ++n_prox) {
                    const double x = -sqrt((double)di * di + (double)dj * dj) / gps[g];
                    if (gm_asu64(exp(x)) != gm_asu64(glibc_exp(x))) {
                        if (bad_exp < 5) printf("exp(%a): libm %a restatement %a\n", x, exp(x), glibc_exp(x));
                        ++bad_exp;
                    }
                }
    }
    if (bad_exp || bad_powf) {
        printf("libm_check FAILED: exp %ld of %ld, powf %ld of %ld differ\n", bad_exp, n, bad_powf, n_powf);
        return 1;
    }
    printf("libm_check ok: exp %ld arguments + %ld proximity-table arguments, powf(x, 1/3) all %ld floats of [0.008856, 1.3], powf(x, 2.4) the 256 sRGB "
           "arguments: bit-identical to this libm\n", n, n_prox, n_powf);
    return 0;
}
