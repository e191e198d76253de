"""The loop of the reference's example/dynamic_obs/dynamic_obs.py (7 moving circles, T = 10, max_obs_num = 6, min_sd = 0.5, wu = 0.2: its own
MPC keywords, dynamic_obs.py:22), run without a simulator window: the headless world replaces `irsim`, `rda_planner_amd.MPC` replaces
`RDA_planner.mpc.MPC` (GPU backend).

    python examples/dynamic_obs_headless.py [world.yaml]
"""
import os
import sys
import time
from collections import namedtuple

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import rda_planner_amd.world as irsim                      # instead of: import irsim
from rda_planner_amd import scenarios as sc
from rda_planner_amd.mpc import MPC                        # instead of: from RDA_planner.mpc import MPC

MPC_KW = dict(receding=10, process_num=5, iter_num=2, max_edge_num=4, max_obs_num=6, min_sd=0.5, wu=0.2, obstacle_order=True)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    world = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "world_dynamic_obs.yaml")
    env = irsim.make(world, save_ani=False, display=False)
    car = namedtuple("car", "G h cone_type wheelbase max_speed max_acce dynamics")
    robot_info = env.get_robot_info()
    car_tuple = car(robot_info.G, robot_info.h, robot_info.cone_type, robot_info.wheelbase, [10, 1], [10, 1.0], "acker")
    ref_path_list = sc.path_track_ref()                    # (the example's dynamic_obs.npy is a byte-identical copy of the path_track reference path)
    mpc_opt = MPC(car_tuple, ref_path_list, sample_time=env.step_time, **MPC_KW)
    t0, steps, min_clear = time.perf_counter(), 0, float("inf")
    for i in range(500):
        obs_list = env.get_obstacle_info_list()
        opt_vel, info = mpc_opt.control(env.robot.state, 6, obs_list)
        env.step(opt_vel)
        env.render(show_traj=True)
        steps += 1
        min_clear = min(min_clear, env.clearance())
        if env.done():
            break
        if info["arrive"]:
            print("arrive at the goal")
            break
    dt = time.perf_counter() - t0
    print(f"{steps} steps, {steps / dt:.0f} steps/s, min clearance {min_clear:.2f} m, collided={env.collided}")


if __name__ == "__main__":
    main()
