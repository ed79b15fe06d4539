/* filter kernels: REPL=1 FUSED=1 K2=1 */
#define HSGPU_INST_REPL true
#define HSGPU_INST_FUSED true
#define HSGPU_INST_K2 true
#define HSGPU_INST_NAME hsgpu_filter_kernels_r1f1k1
#include "scan_inst.inc"
