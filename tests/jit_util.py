"""Helpers of the run-time-scene tests: the text of a scene compiled ahead of time, cut out of
its header and renamed to `Scene`, is a ready-made run-time scene whose pixels are known."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdf_playground_amd", "csrc")
SCENES_DIR = os.path.join(os.path.dirname(CSRC), "scenes")


def aot_scene_source(struct_name, files=("sdfr_scenes.h", "sdfr_scenes2.h", "sdfr_scenes3.h", "sdfr_scenes4.h")):
    for fn in files:
        text = open(os.path.join(CSRC, fn)).read()
        m = re.search(r"^struct %s\n\{\n.*?^\};\n" % struct_name, text, re.S | re.M)
        if m:
            return re.sub(r"\b%s\b" % struct_name, "Scene", m.group(0))
    raise KeyError(struct_name)
