/* the small-batch server's kernels (hwlm_server_kernel = the fused body in a resident workgroup): REPL=1 K2=0 */
#define HSGPU_INST_REPL true
#define HSGPU_INST_FUSED true
#define HSGPU_INST_K2 false
#define HSGPU_INST_NAME_SRV hsgpu_server_kernels_r1f1k0
#include "scan_inst.inc"
