#!/bin/bash
# tools/box_info.sh -- what this GPU box is, recorded with every check / bench run (round 5 saw one box on which every process died with
# "Memory access fault by GPU" before its first scan returned: the next such box should at least be identifiable).
echo "== date"; date -u
echo "== uname"; uname -a
echo "== env"; env | grep -E '^(HSA_|HIP_|ROCR_|GPU_|AMD_|ROCM_|NCCL_|RCCL_)' | sort
echo "== rocminfo (agents)"; /opt/rocm/bin/rocminfo 2>&1 | grep -E 'Agent [0-9]|Marketing Name|  Name:|Node:|Compute Unit|Chip ID|ASIC Revision|Cacheline|Max Waves|Wavefront Size|Workgroup Max Size:|Shader Engines|Shader Arrs|Memory Properties|Internal Node ID|XNACK|Features|Uuid' | head -80
echo "== kfd topology"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n: gpu_id=$(cat $n/gpu_id 2>/dev/null) name=$(cat $n/name 2>/dev/null)"; grep -E 'simd_count|array_count|cu_per_simd_array|max_waves_per_simd|lds_size_in_kb|num_xcc|fw_version|sdma_fw_version|drm_render_minor|hive_id|device_id|unique_id|gfx_target_version|num_sdma' $n/properties 2>/dev/null | tr '\n' ' '; echo; done
echo "== driver"; cat /sys/module/amdgpu/version 2>/dev/null; modinfo amdgpu 2>/dev/null | grep -E '^(version|srcversion)'
echo "== rocm-smi"; /opt/rocm/bin/rocm-smi --showproductname --showdriverversion --showfwinfo --showmemuse --showcomputepartition --showmemorypartition --showclocks --showpower 2>&1 | grep -v '^$' | head -80
echo "== hip"; /opt/rocm/bin/hipconfig --version 2>/dev/null; python - <<'PY'
import torch
print("torch", torch.__version__, "hip", torch.version.hip)
if torch.cuda.is_available():
    p = torch.cuda.get_device_properties(0)
    print(p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2**30, 1), "gcn", getattr(p, "gcnArchName", "?"))
PY
