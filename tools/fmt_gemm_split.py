import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["P"],d["T"],d["C"],d["K"],d.get("affine"),d["native"]["ms"],d["split"]["ms"],d["split"]["tflops"])
