import sys, os, json, hashlib, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
G="tests/golden"; index=json.load(open(G+"/index.json"))
sa.set_piluts(np.fromfile(G+"/piluts_ref.f32",dtype="<f4").reshape(12,2048))
sha=lambda a: hashlib.sha256(a.tobytes()).hexdigest()
t=time.time(); p=sa.Generator(vb.config2(),44100).render(chunk=44100); print("c2",len(p),sha(p)==index["configs"]["config2"]["sha256"],time.time()-t)
t=time.time(); p=sa.Generator(vb.config5(),44100).render(chunk=441000); q=sa.Generator(vb.config5(),44100).render(chunk=11289); print("c5",len(p),len(q),sha(p)==sha(q), sha(p)[:16], time.time()-t)
prgs=[sa.Program.from_image(open(f"{G}/programs/config4_seed{k}.saup","rb").read()) for k in range(4)]
t=time.time(); outs=sa.Batch(prgs,44100).render(chunk=441000); print("c4",[len(o) for o in outs],[sha(o)==index["configs"][f"config4_seed{k}"]["sha256"] for k,o in enumerate(outs)],time.time()-t)
