/* gblastn_amd_kernels.h -- parameter blocks of the gbn_launch_* entry points
 * (plain structs of device pointers and scalars). */
#ifndef GBLASTN_AMD_KERNELS_H
#define GBLASTN_AMD_KERNELS_H
#include "../gblastn_amd/csrc/gbn_dev.h"
#endif
