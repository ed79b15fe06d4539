/* filter kernels: REPL=0 FUSED=0 K2=0 */
#define HSGPU_INST_REPL false
#define HSGPU_INST_FUSED false
#define HSGPU_INST_K2 false
#define HSGPU_INST_NAME hsgpu_filter_kernels_r0f0k0
#include "scan_inst.inc"
