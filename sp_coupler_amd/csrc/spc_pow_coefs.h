/* spc_pow_coefs.h -- the polynomial coefficients of spc_pow (spc_pow.h), as one list */
#ifndef SPC_POW_COEFS
/* The 21 polynomial coefficients: 2/21 ... 2/3 of log m = 2 f + f s P(s), then 1/13! ... 1/3! of exp r = 1 + r + r^2/2 + r^3 q(r) */
#define SPC_POW_COEFS 2.0 / 21.0, 2.0 / 19.0, 2.0 / 17.0, 2.0 / 15.0, 2.0 / 13.0, 2.0 / 11.0, 2.0 / 9.0, 2.0 / 7.0, 2.0 / 5.0, 2.0 / 3.0, 1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0
#endif
