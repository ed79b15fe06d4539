/* the small-batch server's kernels (hwlm_server_kernel = the fused body in a resident workgroup): REPL=0 K2=1 */
#define HSGPU_INST_REPL false
#define HSGPU_INST_FUSED true
#define HSGPU_INST_K2 true
#define HSGPU_INST_NAME_SRV hsgpu_server_kernels_r0f1k1
#include "scan_inst.inc"
