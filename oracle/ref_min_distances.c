/* ORACLE (test infrastructure): exports the reference's own brute-force nearest-point search.  The function body
 * is NOT here: the reference header pytorch-sandbox/generators/utils/calc_min_distances.h is compiled from where it
 * lies under /root/reference (oracle/Makefile passes its directory with -I); this file only gives it a translation
 * unit.  Output: oracle/_ref/libmindist.so (git-ignored, travels to the GPU box like the built product library). */
#include "calc_min_distances.h"
