/* ref_time_shim.h -- TEST INFRASTRUCTURE ONLY.  Force-included (-include) when the reference's
 * Final/photon sources are compiled for oracle/_ref/libref_final.so: the reference seeds its random
 * engine with time(nullptr) at static-initialisation time (photon.h:16); pinning the seed makes its
 * draws repeatable.  No reference file is edited or replaced. */
#include <ctime>
#define RT_REF_FIXED_SEED 20181231
#define time(x) ((time_t)RT_REF_FIXED_SEED)
