"""Measurement switches for the geometry chain -- NOT product code.

Until round 3 these lived inside situation3d_amd/geometry.py behind SIG3D_PROBE_* environment variables, where a
leaked variable could silently skip work inside a captured training step.  The product now has three inert hooks
(GeometryPlan._stop_after, GeometryPipeline._slot_body, GeometryPipeline._skip_chain) and reads no such variable;
`install()` replaces the hooks with the probes below, which read the variables when a step is BUILT:

  SIG3D_PROBE_FPS_ONLY=1              the captured chain stops after the first level's FPS
  SIG3D_PROBE_CHAIN_UNTIL=sampling    ... after FPS + proofs + centre gathers of all levels
  SIG3D_PROBE_CHAIN_UNTIL=ballquery   ... after the ball queries (no distinct-neighbour lists)
  SIG3D_PROBE_SPIN_US=<us>            a slot's graph holds idle workgroups instead of running the chain
  SIG3D_PROBE_SPIN_SHAPE=b,t,vgpr,lds_kb   ... their shape (default 1,64,0,4)
  SIG3D_PROBE_SKIP_CHAIN=1            a slot's graph does nothing (the plan of a real batch stays in place)
  SIG3D_PROBE_SKIP_CHAIN=2            no chain at all, yet every batch gets ITS plan (computed once, cached)

bench.py refuses to run while any SIG3D_PROBE_* variable is set; tools/ab_step.py installs the probes.
"""
import os

import torch


def install():
    from situation3d_amd import _lib, geometry

    def stop_after(self, stage):
        if not torch.cuda.is_current_stream_capturing():
            return False
        if stage == "fps0":
            return os.environ.get("SIG3D_PROBE_FPS_ONLY") == "1"
        return os.environ.get("SIG3D_PROBE_CHAIN_UNTIL") == stage

    def slot_body(self, slot):
        spin = os.environ.get("SIG3D_PROBE_SPIN_US")
        if spin:
            blocks, threads, vg, lds = [int(x) for x in os.environ.get("SIG3D_PROBE_SPIN_SHAPE", "1,64,0,4").split(",")]
            if not hasattr(self, "_probe_sink"):
                self._probe_sink = torch.zeros(16, dtype=torch.float32, device=self.device)
            _lib.call("sig3d_hold", _lib.ptr(self._probe_sink), blocks, threads, int(spin), vg, lds,
                      _lib.stream_ptr(self.device))
        elif os.environ.get("SIG3D_PROBE_SKIP_CHAIN") == "1":
            pass
        else:
            slot["plan"].compute(slot["xyz"])

    def skip_chain(self, point_clouds):
        if os.environ.get("SIG3D_PROBE_SKIP_CHAIN") != "2":
            return False
        cache = self.__dict__.setdefault("_probe_plans", {})
        key = id(point_clouds)
        if key not in cache:
            plan = geometry.GeometryPlan(self.plan_cur.batch, self.plan_cur.n_points, self.plan_cur.levels, self.device)
            cache[key] = (plan.compute(point_clouds[..., :3].contiguous()), point_clouds)
        self.plan_cur.copy_from(cache[key][0])
        return True

    geometry.GeometryPlan._stop_after = stop_after
    geometry.GeometryPipeline._slot_body = slot_body
    geometry.GeometryPipeline._skip_chain = skip_chain
