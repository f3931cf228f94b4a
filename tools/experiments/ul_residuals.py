import numpy as np, torch, sys, os
sys.path.insert(0, "/root/repo")
from nmma_amd import synthetic as syn
from nmma_amd.engine import EMEngine
case = syn.config2_case()
case["theta"] = syn.draw_theta(1000, 4096, case["names"])[1]
eng = EMEngine.from_case(case)
th = torch.as_tensor(case["theta"], device="cuda:0")
tobs, mag = eng.lightcurves(th)
tobs, mag = tobs.cpu().numpy(), mag.cpu().numpy()
times, mags, sigmas = case["data"]
for k, f in enumerate(case["observed_filters"]):
    ul = ~np.isfinite(sigmas[f])
    if ul.any():
        m = case["model_filters"].index(f)
        for t_u, m_u in zip(np.asarray(times[f])[ul], np.asarray(mags[f])[ul]):
            est = np.array([np.interp(t_u, tobs[b], mag[b, m], left=np.inf, right=np.inf) for b in range(len(tobs))])
            sys_sigma = 1.0
            b = (est - m_u) / sys_sigma
            print(f, "UL at", t_u, m_u, "b quantiles", np.nanquantile(b[np.isfinite(b)], [0, 0.05, 0.25, 0.5, 0.75, 0.95, 1]), "frac b<-1:", np.mean(b < -1), "frac b>=8.5:", np.mean(b >= 8.5))
