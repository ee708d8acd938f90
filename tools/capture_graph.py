"""Extract the reference's saved training graph into a fixture: tests/golden/graph.json.

/root/reference/policy_value_loss.json is the symbol dump the reference writes while it builds its
training module (policy_value_net_mxnet.py:194; MXNet 1.5.1): the only machine-readable pin of the
network architecture the reference holds.  This script copies the DATA out of it in a
normalised form -- node list (operator, name, attributes, input wiring), argument nodes and heads --
and drops what is serialisation detail (node_row_ptr, the version stamp is kept as a number).
Run in the build container only (the reference tree does not exist on the GPU box).

  python tools/capture_graph.py [/root/reference/policy_value_loss.json] [tests/golden/graph.json]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/policy_value_loss.json"
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "tests", "golden", "graph.json")
    with open(src) as f:
        g = json.load(f)
    out = {
        "source": "policy_value_loss.json (reference repo root), written by policy_value_net_mxnet.py:194",
        "mxnet_version": g["attrs"]["mxnet_version"][1],
        "nodes": [{"op": n["op"], "name": n["name"], "attrs": n.get("attrs", {}), "inputs": n["inputs"]} for n in g["nodes"]],
        "arg_nodes": g["arg_nodes"],
        "heads": g["heads"],
    }
    with open(dst, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")
    print("wrote %s: %d nodes, %d arguments, %d heads" % (dst, len(out["nodes"]), len(out["arg_nodes"]), len(out["heads"])))


if __name__ == "__main__":
    main()
