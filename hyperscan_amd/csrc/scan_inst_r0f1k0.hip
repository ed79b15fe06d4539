/* filter kernels: REPL=0 FUSED=1 K2=0 */
#define HSGPU_INST_REPL false
#define HSGPU_INST_FUSED true
#define HSGPU_INST_K2 false
#define HSGPU_INST_NAME hsgpu_filter_kernels_r0f1k0
#include "scan_inst.inc"
