/* filter kernels: REPL=0 FUSED=0 K2=1 */
#define HSGPU_INST_REPL false
#define HSGPU_INST_FUSED false
#define HSGPU_INST_K2 true
#define HSGPU_INST_NAME hsgpu_filter_kernels_r0f0k1
#include "scan_inst.inc"
