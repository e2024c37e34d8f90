// shw_ssw_grad2_m32.hip -- the masked 32-keys-per-lane form of the two-wave training kernel, built WITH SLP vectorisation
// (see the note above launch_forward_grad2_masked32 in shw_ssw_grad2.hip, and SLP_UNITS in the Makefile).
#define SHW_GRAD2_MASKED32_UNIT
#include "shw_ssw_grad2.hip"
