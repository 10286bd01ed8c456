#!/usr/bin/env python3
"""Generate tests/golden/student_graph_<ckpt>.json from the reference's TF1 MetaGraphDef.

Runs ONLY in the build container (reads /root/reference, which never travels to the
GPU box).  The output is data: variable names/shapes, the trainable order, and the
attributes of the compute nodes of checkpoints/<ckpt>/model.meta.  tests/test_spec.py
pins ams_amd/spec.py (the architecture the HIP engine and the oracle are built from)
against it.

No TensorFlow/protobuf needed: a minimal protobuf wire-format reader (varint +
length-delimited) is enough for MetaGraphDef → GraphDef → NodeDef → AttrValue.
"""
import json
import struct
import sys
from pathlib import Path

REF = Path("/root/reference/checkpoints")
OUT = Path(__file__).resolve().parent


def _varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def fields(buf):
    """Yield (field_number, wire_type, value) for one protobuf message body."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, v


def parse_shape(buf):
    dims = []
    for fno, _, v in fields(buf):
        if fno == 2:  # dim
            size = 0
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return dims


def parse_tensor(buf):
    out = {"dtype": None, "shape": [], "floats": [], "ints": [], "content": b""}
    for fno, wt, v in fields(buf):
        if fno == 1:
            out["dtype"] = v
        elif fno == 2:
            out["shape"] = parse_shape(v)
        elif fno == 4:
            out["content"] = bytes(v)
        elif fno == 5:
            if wt == 5:
                out["floats"].append(struct.unpack("<f", v)[0])
            else:
                out["floats"].extend(struct.unpack("<%df" % (len(v) // 4), v))
        elif fno == 7:
            if wt == 0:
                out["ints"].append(v)
            else:
                p = 0
                while p < len(v):
                    x, p = _varint(v, p)
                    out["ints"].append(x)
    if out["content"] and out["dtype"] == 1:
        out["floats"] = list(struct.unpack("<%df" % (len(out["content"]) // 4), out["content"]))
    if out["content"] and out["dtype"] == 3:
        out["ints"] = list(struct.unpack("<%di" % (len(out["content"]) // 4), out["content"]))
    return out


def parse_attr(buf):
    for fno, wt, v in fields(buf):
        if fno == 2:
            return bytes(v).decode("utf8", "replace")
        if fno == 3:
            return v if v < (1 << 63) else v - (1 << 64)
        if fno == 4:
            return struct.unpack("<f", v)[0]
        if fno == 5:
            return bool(v)
        if fno == 6:
            return ("type", v)
        if fno == 7:
            return ("shape", parse_shape(v))
        if fno == 8:
            return ("tensor", parse_tensor(v))
        if fno == 1:  # list
            ints, strs = [], []
            for f2, w2, v2 in fields(v):
                if f2 == 3:
                    if w2 == 0:
                        ints.append(v2)
                    else:
                        p = 0
                        while p < len(v2):
                            x, p = _varint(v2, p)
                            ints.append(x)
                elif f2 == 2:
                    strs.append(bytes(v2).decode())
            return ints if ints else strs
    return None


def parse_node(buf):
    node = {"name": "", "op": "", "input": [], "attr": {}}
    for fno, _, v in fields(buf):
        if fno == 1:
            node["name"] = bytes(v).decode()
        elif fno == 2:
            node["op"] = bytes(v).decode()
        elif fno == 3:
            node["input"].append(bytes(v).decode())
        elif fno == 5:
            k = val = None
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    k = bytes(v2).decode()
                elif f2 == 2:
                    val = parse_attr(v2)
            node["attr"][k] = val
    return node


def parse_meta(path):
    buf = memoryview(path.read_bytes())
    nodes = []
    collections = {}
    info = {}
    for fno, _, v in fields(buf):
        if fno == 1:  # meta_info_def
            for f2, _, v2 in fields(v):
                if f2 == 5:
                    info["tf_version"] = bytes(v2).decode()
        elif fno == 2:  # graph_def
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    nodes.append(parse_node(v2))
        elif fno == 4:  # collection_def map entry
            key = None
            names = []
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    key = bytes(v2).decode()
                elif f2 == 2:  # CollectionDef
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:  # node_list
                            for f4, _, v4 in fields(v3):
                                if f4 == 1:
                                    names.append(bytes(v4).decode())
                        elif f3 == 2:  # bytes_list of serialized VariableDef
                            for f4, _, v4 in fields(v3):
                                if f4 == 1:
                                    for f5, _, v5 in fields(v4):
                                        if f5 == 1:
                                            names.append(bytes(v5).decode())
            collections[key] = names
    return info, nodes, collections


def const_value(by_name, name):
    """Follow Identity chains to a Const and return its tensor payload."""
    name = name.split(":")[0].lstrip("^")
    seen = 0
    while seen < 8:
        n = by_name[name]
        if n["op"] == "Const":
            t = n["attr"]["value"][1]
            return t["floats"] or t["ints"]
        if n["op"] in ("Identity",):
            name = n["input"][0].split(":")[0]
            seen += 1
            continue
        return None
    return None


def build(ckpt):
    info, nodes, coll = parse_meta(REF / ckpt / "model.meta")
    by_name = {n["name"]: n for n in nodes}
    variables = []
    for n in nodes:
        if n["op"] == "VariableV2":
            variables.append({"name": n["name"] + ":0", "shape": n["attr"]["shape"][1]})
    compute = []
    for n in nodes:
        if n["op"] in ("Conv2D", "DepthwiseConv2dNative"):
            compute.append({"name": n["name"], "op": n["op"], "strides": n["attr"].get("strides"),
                            "dilations": n["attr"].get("dilations"), "padding": n["attr"].get("padding"),
                            "inputs": n["input"]})
        elif n["op"] == "FusedBatchNormV3":
            compute.append({"name": n["name"], "op": n["op"], "epsilon": n["attr"].get("epsilon"),
                            "is_training": n["attr"].get("is_training"), "inputs": n["input"]})
        elif n["op"] in ("SpaceToBatchND", "BatchToSpaceND"):
            compute.append({"name": n["name"], "op": n["op"], "inputs": n["input"],
                            "block_shape": const_value(by_name, n["input"][1]),
                            "pad_or_crop": const_value(by_name, n["input"][2])})
        elif n["op"] in ("Relu6", "Relu", "AddV2", "Mean", "BiasAdd", "ConcatV2", "ResizeBilinear",
                         "Mul", "Sub", "PadV2", "FIFOQueueV2", "StopGradient"):
            entry = {"name": n["name"], "op": n["op"], "inputs": n["input"]}
            if n["op"] == "ResizeBilinear":
                entry["align_corners"] = n["attr"].get("align_corners")
                entry["half_pixel_centers"] = n["attr"].get("half_pixel_centers")
            if n["op"] == "Mean":
                entry["keep_dims"] = n["attr"].get("keep_dims")
                entry["axes"] = const_value(by_name, n["input"][1])
            if n["op"] == "FIFOQueueV2":
                entry["capacity"] = n["attr"].get("capacity")
            if n["op"] in ("Mul", "Sub") and "/" not in n["name"]:
                entry["const"] = [const_value(by_name, i) for i in n["input"]]
            if n["op"] == "ConcatV2" and "/" not in n["name"]:
                entry["const"] = [const_value(by_name, i) for i in n["input"]]
            if n["op"] in ("Mul", "Sub", "AddV2") and ("BatchNorm" in n["name"] or "Initializer" in n["name"]
                                                      or n["name"].startswith("ones")):
                continue
            compute.append(entry)
    ema = {}
    for n in nodes:
        if n["op"] == "AssignSub" and "AssignMovingAvg" in n["name"]:
            # AssignSub(var, (var - batch_stat) * (1 - decay));  decay = <scope>/BatchNorm/Const_2
            mul = by_name[n["input"][1].split(":")[0]]
            sub_stat = by_name[mul["input"][0].split(":")[0]]
            sub_decay = by_name[mul["input"][1].split(":")[0]]
            ema[n["name"]] = {"var": n["input"][0], "stat": sub_stat["input"][1],
                              "one": const_value(by_name, sub_decay["input"][0]),
                              "decay": const_value(by_name, sub_decay["input"][1])}
    out = {
        "source": "checkpoints/%s/model.meta" % ckpt,
        "tf_version": info.get("tf_version"),
        "n_nodes": len(nodes),
        "variables": variables,
        "trainable_variables": coll.get("trainable_variables", []),
        "model_variables": coll.get("model_variables", []),
        "update_ops": coll.get("update_ops", []),
        "compute_nodes": compute,
        "ema": ema,
    }
    return out


if __name__ == "__main__":
    for ckpt, tag in (("deeplabv3_mobilenetv2_cityscapes", "cityscapes"),
                      ("deeplabv3_mobilenetv2_pascalvoc2012", "pascalvoc2012")):
        data = build(ckpt)
        path = OUT / ("student_graph_%s.json" % tag)
        path.write_text(json.dumps(data, indent=0, sort_keys=True))
        print(path, len(data["variables"]), "variables,", len(data["trainable_variables"]), "trainable,",
              len(data["compute_nodes"]), "compute nodes")
