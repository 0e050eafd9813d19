"""Worker of tests/test_gpu_multistart.py::test_suggestion_on_N_ranks_* (launched under torch.distributed.run, one process per rank,
all ranks on the box's one GPU; gloo carries the exchange): the whole-suggestion drivers of cornell_moe_amd/multistart.py with
comm = dist.Exchange -- restarts dealt to the ranks (single GP), members dealt to the ranks (MCMC ensemble) -- next to the
single-rank answers computed by the same process.  Writes <out>.<rank>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from cornell_moe_amd import GPP, api, multistart
    from cornell_moe_amd import dist as mdist
    from helpers import load_golden_kg_multistart
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = mdist.Exchange(rank, world)
    res = {}
    try:
        cases, mcmc = load_golden_kg_multistart()

        def params(gd, domain_type=GPP.DomainTypes.tensor_product):
            class P(object):
                optimizer_type = GPP.OptimizerTypes.gradient_descent
                num_random_samples = 0
            P.domain_type = domain_type
            P.optimizer_parameters = GPP.GradientDescentParameters(
                *[t(v) for t, v in zip((int, int, int, int, float, float, float, float), gd)])
            return P()

        def rnd():
            r = GPP.RandomnessSourceContainer(1)
            r.SetExplicitUniformGeneratorSeed(314)
            r.SetExplicitNormalRNGSeed(271)
            return r

        # ---- single GP, restarts dealt to the ranks: the reference's fixtures, starts from the fixture (same on every rank) ----
        for c in cases[:2]:
            i = c.inp
            f, M = int(i["num_fidelity"]), int(i["M"])
            G = api.DeviceGP(np.concatenate([[float(i["alpha"])], i["lengths"]]), i["X"], i["y"], i["noise"], list(i["derivs"]))
            Xp = i["Xp"] if int(i["p"]) > 0 else None
            args = (tuple(i["outer_gd"]), tuple(i["inner_gd"]), i["bounds"], i["discrete"], i["starts"], Xp, M, float(i["best_so_far"]),
                    i["normals"])
            one = G.kg_multistart(*args, num_fidelity=f)
            many = G.kg_multistart(*args, num_fidelity=f, comm=ex)
            res["kg%d_one" % c.index], res["kg%d_many" % c.index] = np.r_[one[0].ravel(), one[1], one[2]], np.r_[many[0].ravel(), many[1], many[2]]
            res["kg%d_ref" % c.index] = c.out["best_point"].ravel()
            # the drop-in driver with Latin-hypercube starts drawn on every rank from the same seeds
            a = multistart.kg_optimal_points(G, f, params(i["outer_gd"]), params(i["inner_gd"]), i["bounds"], i["discrete"], Xp, int(i["q"]),
                                             float(i["best_so_far"]), M, rnd())
            b = multistart.kg_optimal_points(G, f, params(i["outer_gd"]), params(i["inner_gd"]), i["bounds"], i["discrete"], Xp, int(i["q"]),
                                             float(i["best_so_far"]), M, rnd(), comm=ex)
            res["opt%d_one" % c.index], res["opt%d_many" % c.index] = np.r_[a[0].ravel(), a[1]], np.r_[b[0].ravel(), b[1]]
        # ---- MCMC ensemble, members dealt to the ranks (world <= number of members) ----
        for mk in mcmc:
            i = mk.inp
            nm = i["hypers"].shape[0]
            if world > nm:
                continue
            f, M = int(i["num_fidelity"]), int(i["M"])
            Xp = i["Xp"] if int(i["p"]) > 0 else None
            args = (tuple(i["outer_gd"]), tuple(i["inner_gd"]), i["bounds"], i["discrete"], i["starts"], Xp, M, i["best_so_far"], i["normals"])
            whole = api.DeviceGPMCMC(i["hypers"], i["noises"], i["X"], i["y"], ())
            one = whole.kg_multistart(*args, num_fidelity=f)
            mine = api.DeviceGPMCMC(i["hypers"], i["noises"], i["X"], i["y"], (), members=mdist.shard_members(nm, rank, world))
            many = mine.kg_multistart(*args, num_fidelity=f, comm=ex)
            res["mc%d_one" % mk.index], res["mc%d_many" % mk.index] = np.r_[one[0].ravel(), one[1], one[2]], np.r_[many[0].ravel(), many[1], many[2]]
            res["mc%d_ref" % mk.index] = mk.out["best_point"].ravel()
        res["exchange"] = np.array([ex.calls, ex.doubles, ex.seconds])
        np.savez(out + ".%d.npz" % rank, **res)
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
