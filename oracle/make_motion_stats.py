#!/usr/bin/env python
"""Write ``artalk_amd/assets/motion_stats.json`` from the reference's constant table (build container only).

``BITWISE_VAE`` and ``StyleEncoder`` register ``ALLTALKEMICA_MEAN/STD`` (``/root/reference/app/modules/data_stats.py:1-32``)
as the persistent buffers ``motion_mean`` / ``motion_std`` (``bitwise_vae.py:24-25``, ``style_encoder.py:12-13``), so the
212 numbers are part of every checkpoint.  No real checkpoint exists offline; the synthetic-weight generator
(``artalk_amd/weights.py``) therefore takes these buffers from this data file instead of inventing stand-ins, so the
goldens exercise the reference's real statistics (dims 100:103 have mean 0, std down to 0.0235).

Only the numbers are extracted (the module is a pure table of float literals).  Usage: python oracle/make_motion_stats.py
"""
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/app/modules/data_stats.py"


def main():
    spec = importlib.util.spec_from_file_location("_ref_data_stats", SRC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean, std = [float(x) for x in mod.ALLTALKEMICA_MEAN], [float(x) for x in mod.ALLTALKEMICA_STD]
    assert len(mean) == len(std) == 106 and mean[100:103] == [0.0, 0.0, 0.0] and min(std) > 0
    out = {"_comment": "motion_mean / motion_std buffers of the reference checkpoint (ALLTALKEMICA statistics, 106 dims); "
                       "written by oracle/make_motion_stats.py", "motion_mean": mean, "motion_std": std}
    path = os.path.join(REPO, "artalk_amd", "assets", "motion_stats.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
