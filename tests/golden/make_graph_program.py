#!/usr/bin/env python3
"""Generate tests/golden/student_program_<ckpt>.json: the forward sub-graph of the reference's MetaGraphDef as an
executable node list.

Runs ONLY in the build container (reads /root/reference/checkpoints/<ckpt>/model.meta through the wire-format reader
of make_graph_fixture.py; no TensorFlow).  The output is DATA decoded from the reference's own file: every ancestor node
of the last ResizeBilinear (= `student_logits`, reference utils/graph_utils.py:353-358) in topological order, with its op,
its inputs (producer name + output index), the attributes an executor needs, and the payload of every Const.  Variables
appear as `VariableV2` leaves (fed from the `.npy` weight dict by name), the dequeued frame batch as the `features` node.

`oracle/graph_interp.py` executes this list op by op.  That makes the WIRING of the CPU oracle (which layer feeds which,
strides, paddings, eps, residual adds, head) the reference's, not a table this repository wrote: `ams_amd/spec.py`, the
two hand-written oracles and the HIP engine are all compared with it.
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_graph_fixture as M  # noqa: E402

KEEP_ATTRS = ("strides", "padding", "dilations", "epsilon", "is_training", "axis", "N", "keep_dims", "align_corners",
              "half_pixel_centers", "DstT", "SrcT", "begin_mask", "end_mask", "ellipsis_mask", "new_axis_mask",
              "shrink_axis_mask", "data_format")


def _split(inp):
    name, _, idx = inp.partition(":")
    return name, int(idx or 0)


def build(ckpt):
    _info, nodes, _coll = M.parse_meta(M.REF / ckpt / "model.meta")
    by_name = {n["name"]: n for n in nodes}
    out_name = [n["name"] for n in nodes if n["op"] == "ResizeBilinear"][-1]
    order, state = [], {}

    def visit(name):
        stack = [(name, False)]
        while stack:
            nm, done = stack.pop()
            if done:
                order.append(nm)
                state[nm] = 2
                continue
            if state.get(nm):
                continue
            state[nm] = 1
            stack.append((nm, True))
            n = by_name[nm]
            if n["op"] in ("VariableV2", "QueueDequeueV2"):
                continue
            for i in reversed(n["input"]):
                assert not i.startswith("^"), "control input in the forward graph: %s" % i
                if not state.get(_split(i)[0]):
                    stack.append((_split(i)[0], False))

    visit(out_name)
    program = []
    for nm in order:
        n = by_name[nm]
        e = {"name": nm, "op": n["op"]}
        if n["op"] not in ("VariableV2", "QueueDequeueV2"):
            e["inputs"] = [list(_split(i)) for i in n["input"]]
        for k in KEEP_ATTRS:
            if k in n["attr"] and n["attr"][k] is not None:
                v = n["attr"][k]
                e[k] = v[1] if isinstance(v, tuple) else v
        if n["op"] == "Const":
            t = n["attr"]["value"][1]
            e["dtype"] = t["dtype"]
            e["shape"] = t["shape"]
            e["value"] = t["floats"] if t["dtype"] == 1 else t["ints"]
        if n["op"] == "VariableV2":
            e["shape"] = n["attr"]["shape"][1]
        program.append(e)
    return {"source": "checkpoints/%s/model.meta" % ckpt, "output": out_name, "feed": "features", "nodes": program}


if __name__ == "__main__":
    for ckpt, tag in (("deeplabv3_mobilenetv2_cityscapes", "cityscapes"), ("deeplabv3_mobilenetv2_pascalvoc2012", "pascalvoc2012")):
        data = build(ckpt)
        path = M.OUT / ("student_program_%s.json" % tag)
        path.write_text(json.dumps(data, separators=(",", ":"), sort_keys=True))
        print(path, len(data["nodes"]), "nodes,", path.stat().st_size, "bytes")
