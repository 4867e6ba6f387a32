"""Identify the GPU box at the top of every evidence log (VERDICT r5 next #1a): compute / memory partition, driver and firmware
versions, the HIP device properties — so that a box that misbehaves (profiles/r5_gpu_tests_run_with_9_failures.txt) can be told
from its neighbours after the fact.  `python tools/box_info.py` prints one JSON object; bench.py and tests/conftest.py call box_info()."""
import json
import re
import shutil
import subprocess


def _run(cmd, timeout=20):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout).stdout
    except Exception as e:  # noqa: BLE001 - identification only, never fatal
        return f"<{type(e).__name__}: {e}>"


def box_info(device=0):
    info = {}
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(device)
            info.update(name=p.name, gcn_arch=getattr(p, "gcnArchName", None), compute_units=p.multi_processor_count,
                        hbm_gb=round(p.total_memory / 2 ** 30, 1), torch=torch.__version__, hip=torch.version.hip)
    except Exception as e:  # noqa: BLE001
        info["torch_error"] = str(e)
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    out = _run([smi, "--showcomputepartition", "--showmemorypartition", "--showdriverversion", "--showfwinfo", "--showuniqueid", "--json"])
    try:
        j = json.loads(out[out.index("{"):])
        card = j.get(f"card{device}", {})
        info["compute_partition"] = card.get("Compute Partition")
        info["memory_partition"] = card.get("Memory Partition")
        info["unique_id"] = card.get("Unique ID")
        info["driver"] = (j.get("system") or {}).get("Driver version")
        info["firmware"] = {k.replace(" firmware version", ""): v for k, v in card.items() if "firmware version" in k}
    except Exception:  # noqa: BLE001 - older rocm-smi: keep the raw text
        info["rocm_smi_raw"] = out[-1500:]
    ri = _run([shutil.which("rocminfo") or "/opt/rocm/bin/rocminfo"])
    m = re.search(r"gfx950[\s\S]*?Compute Unit:\s*(\d+)[\s\S]*?(?:Num XCC|XCC):\s*(\d+)", ri) if ri else None
    if m:
        info["rocminfo_cu"], info["xcc"] = int(m.group(1)), int(m.group(2))
    else:
        x = re.search(r"(?:Num XCC|XCC)\w*:\s*(\d+)", ri or "")
        if x:
            info["xcc"] = int(x.group(1))
    return info


if __name__ == "__main__":
    print(json.dumps(box_info()))
