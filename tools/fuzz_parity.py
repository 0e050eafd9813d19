"""Randomised parity sweep on the GPU: random shapes / kernels / derivative sets / fidelity dimensions / optimiser settings,
device q-KG and q-EI against the unmodified reference (oracle/_ref, when built; the plain-C restatement otherwise) on the same normal tables.  Prints every violation of the stated tolerances.
    python tools/fuzz_parity.py [num_cases] [seed] [n_max = 300] [d_max = 16] [g_max = 4]
(n_max > 256 reaches the wave-per-sample kernel's many-tile instantiation and its multi-trial passes; d_max up to 32 and g_max up to
12 reach the wide-dimension and 8 / 12-slot instantiations)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cornell_moe_amd import api  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402
from helpers import TOL, reference_checker  # noqa: E402
from oracle import orc  # noqa: E402



def run(num_cases=100, seed=2026, n_max=300, d_max=16, g_max=4):
  rng = np.random.default_rng(seed)
  bad = 0
  for case in range(num_cases):
      d = int(rng.integers(1, d_max + 1))
      g = int(rng.integers(0, min(g_max, d) + 1)) if rng.uniform() < 0.5 else 0
      derivs = tuple(int(v) for v in rng.permutation(d)[:g])
      umax = (64 if g_max <= 4 else 128) // (1 + g)
      q = int(rng.integers(1, min(4, umax) + 1))
      p = int(rng.integers(0, min(3, umax - q) + 1))
      n = int(rng.integers(1, n_max))
      P = int(rng.integers(1, 13))
      M = int(rng.integers(1, 65))
      cov = int(rng.integers(0, 2))
      f = int(rng.integers(0, d)) if rng.uniform() < 0.3 else 0
      gd = (1, int(rng.integers(1, 8)), int(rng.integers(1, 3)), 3, float(rng.choice([0.0, 0.5, 1.0])),
            float(rng.choice([1.0, 0.3, 2.0])), float(rng.choice([0.1, 0.5, 1.0])), float(rng.choice([1e-10, 1e-6])))
      w = make_workload(seed=10_000 + case, n=n, d=d, q=q, M=M, P=P, derivs=derivs, p=p)
      aff = ""
      if rng.uniform() < 0.4:  # the same problem in an offset / rescaled domain: x' = shift + scale x, l' = scale l
          shift = rng.choice([-50.0, 10.0, 100.0, 1000.0], size=d) * (rng.uniform(size=d) < 0.7)
          scale = rng.choice([0.1, 1.0, 10.0], size=d)
          for name in ("X", "Xq", "Xp", "discrete"):
              setattr(w, name, shift + scale * getattr(w, name))
          w.lengths = w.lengths * scale
          w.hyperparameters = np.concatenate([[w.alpha], w.lengths])
          w.bounds = np.column_stack([shift, shift + scale]).reshape(-1)
          aff = " affine(shift max %g)" % np.abs(shift).max()
      disc = w.discrete[:, :d - f]
      bounds = w.bounds[:2 * (d - f)]
      try:
          O = orc.OrcGP(cov, w.alpha, w.lengths, w.X, w.y, w.noise, derivs)
      except Exception:
          continue
      G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, derivs, cov_type=cov)
      full = np.hstack([disc, np.ones((disc.shape[0], f))])
      best = float(O.additional_mean(full).min())
      Xp = w.Xp if p else None
      ro = O.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f)
      # (r4) the unmodified reference as the checker of KG / grad KG / end points where oracle/_ref is built; the restatement keeps
      # the pass counts
      R = reference_checker(cov, w.alpha, w.lengths, w.X, w.y, w.noise, derivs)
      rc = R.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f) if R is not None else ro
      scale = max(float(np.abs(rc["grad"]).max()), abs(rc["kg"]), 1e-6)
      for variant in ("0", "1", "2"):
          os.environ["MOE_KG_VARIANT"] = variant
          try:
              rg = G.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f, want_best_points=True)
          except api.OptimalLearningException as e:
              # a FORCED kernel that cannot hold the point set (d > 16: all tiles in LDS only) or is not built for the shape (the
              # streamed-weights kernel: every derivative slot observed, <= 4 of them) refuses loudly -- not a parity case
              # (r4: "too large" under any forced variant -- shapes with 5..7 / 9..11 observed derivatives keep to the workgroup-per-sample
              #  kernel whatever is asked, and it has its size limit)
              if "too large" in str(e) or (variant == "2" and "streamed-weights" in str(e)):
                  continue
              raise
          if int(variant) != G.last_kernel_info()["variant"]:
              continue  # (a shape only one kernel is built for keeps it whatever is asked)
          # beyond the production depth of the inner optimiser (6 steps x 1 restart) samples reach stationary points where
          # accept / restart decisions hinge on differences below rounding: two correct FP64 implementations -- the
          # restatement and the reference itself -- then differ by ~1e-7 on x* and grad KG (tests/helpers.py kg_tolerances;
          # seed 11 case 58: oracle vs oracle/_ref 2.4e-7), and the number of gradient passes is not comparable
          loose = gd[1] * gd[2] > 8
          ptol = 1e-6 if loose else 1e-8
          gtol = 1e-6 if loose else TOL["grad_kg"]
          ktol = TOL["kg"]
          if R is not None:
              # (r4, with the reference as the checker) the same effect at production depth, measured instead of assumed: where the
              # restatement and the reference THEMSELVES differ beyond the tight tolerance (case 71 of seed 31337: one free dimension,
              # SE kernel, max_relative_change = 1 -- 6 of 18 end points differ by up to 9e-8 between the two CPU codes), the device
              # is held to ten times that distance from the reference (the three device kernels round differently from each other, too)
              dis_p = float(np.abs(ro["best_point"] - rc["best_point"]).max())
              dis_g = float(np.abs(ro["grad"] - rc["grad"]).max()) / scale
              dis_k = abs(ro["kg"] - rc["kg"]) / max(abs(rc["kg"]), 1e-6)
              ptol, gtol, ktol = max(ptol, 10.0 * dis_p), max(gtol, 10.0 * dis_g), max(ktol, 10.0 * dis_k)
              # (and where the two CPU codes end a sample 1e-9 or more apart, one of them took a step the other did not: the
              #  restatement's pass count is then not the reference's -- seed 2024 case 127, one free dimension again: 1.0e-8 between
              #  the CPU codes, the device within 3e-9 of the restatement with one gradient pass more)
              loose = loose or dis_p > 1.0e-9
          dev_d = np.abs(rg["best_point"] - rc["best_point"]).max(axis=1)
          e_kg = abs(rg["kg"] - rc["kg"]) / max(abs(rc["kg"]), 1e-6)
          e_gr = float(np.abs(rg["grad"] - rc["grad"]).max()) / scale
          if R is not None and gd[1] * gd[2] <= 8:
              # r6: at production depth with the reference as the checker an end point may differ from the
              # reference's by more than 1e-8 ONLY on a sample where the two CPU codes -- the reference and its restatement --
              # themselves disagree (a decision of that sample's line search hinges on a difference below rounding); any other
              # mismatch fails the case.  Such a flipped sample moves grad KG by ~ 1e-5 / M: that much, per flipped sample, is
              # granted -- nothing for samples that agree.
              cpu_d = np.abs(ro["best_point"] - rc["best_point"]).max(axis=1)
              knife = cpu_d > 1.0e-9
              flipped = knife & (dev_d > 1.0e-8)
              mism = float(((dev_d > 1.0e-8) & ~knife).mean())
              gtol = max(TOL["grad_kg"], 10.0 * dis_g, 2.0e-5 * float(flipped.sum()) / M)
              ktol = max(TOL["kg"], 10.0 * dis_k, 2.0e-5 * float(flipped.sum()) / M)
              mism_cap = 0.0
          else:
              # (beyond production depth -- or without the reference on the box -- samples reach stationary points and a handful may sit
              #  on decision boundaries: the r4 rule)
              mism = float((dev_d > ptol).mean())
              gtol = max(gtol, 2.0e-5 * mism)
              mism_cap = max(0.05, 2.5 / M)
          if e_kg > ktol or e_gr > gtol or mism > mism_cap or (not loose and rg["grad_evals"] != ro["grad_evals"]):
              bad += 1
              print("KG MISMATCH" + aff + " case %d variant %s: n=%d d=%d q=%d p=%d g=%s f=%d P=%d M=%d cov=%d gd=%s: rel kg %.2e grad %.2e "
                    "best-point mismatch %.3f grad passes %d vs %d" % (case, variant, n, d, q, p, derivs, f, P, M, cov, gd, e_kg,
                                                                    e_gr, mism, rg["grad_evals"], ro["grad_evals"]), flush=True)
      os.environ.pop("MOE_KG_VARIANT", None)
      if q + p <= 16:
          eb = float(np.median(w.y[:, 0]))
          eo, go = R.ei(w.Xq, Xp, M, eb, w.ei_normals)[:2] if R is not None else O.ei(w.Xq, Xp, M, eb, w.ei_normals)
          eg, gg = G.ei(w.Xq, Xp, M, eb, w.ei_normals)
          if abs(eo - eg) > TOL["ei"] * max(abs(eo), 1e-3) or \
                  np.abs(gg - go).max() > TOL["grad_ei"] * max(np.abs(go).max(), abs(eo), 1e-3):
              bad += 1
              print("EI MISMATCH case %d: n=%d d=%d q=%d p=%d g=%s cov=%d: %.3e vs %.3e, grad err %.2e" % (
                  case, n, d, q, p, derivs, cov, eg, eo, float(np.abs(gg - go).max())), flush=True)
  from oracle import ref
  print("fuzz: %d cases, %d violations (checker: %s)" % (num_cases, bad, "oracle/_ref, the unmodified reference" if ref.available()
                                                          else "the C restatement"))
  return bad


if __name__ == "__main__":
  a = [int(v) for v in sys.argv[1:]]
  sys.exit(1 if run(*a) else 0)

