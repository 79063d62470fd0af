/* TEST INFRASTRUCTURE ONLY.  One-off exhaustive pin of the shared sincos (spath_oracle.c) against
 * the libm of the machine it runs on:  gcc -O2 -ffp-contract=off sincos_exhaustive.c liboracle.so -lm
 * Result in the build container (glibc 2.35): 0 mismatches over all floats in [0, 8]. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "spath_oracle.h"
int main(void) {
	uint32_t hi; const float top = 8.0f; memcpy(&hi, &top, 4);
	long bad_s = 0, bad_c = 0, n = 0;
	for (uint32_t u = 0; u <= hi; ++u, ++n) {
		float x; memcpy(&x, &u, 4);
		const float a = spo_sinf(x), b = sinf(x), c = spo_cosf(x), d = cosf(x);
		if (memcmp(&a, &b, 4)) { if (bad_s < 5) printf("sin x=%a oracle=%a libm=%a\n", x, a, b); bad_s++; }
		if (memcmp(&c, &d, 4)) { if (bad_c < 5) printf("cos x=%a oracle=%a libm=%a\n", x, c, d); bad_c++; }
	}
	printf("checked %ld floats in [0,8]: sin mismatches %ld, cos mismatches %ld\n", n, bad_s, bad_c);
	return (bad_s || bad_c) ? 1 : 0;
}
