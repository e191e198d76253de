"""The loop of the reference's example/lidar_nav/lidar_path_track.py, run without a simulator window: the headless world
(with its ray-cast lidar) replaces `irsim`, `rda_planner_amd.lidar.scan_box` replaces the script's DBSCAN + cv2 `scan_box`,
`rda_planner_amd.MPC` replaces `RDA_planner.mpc.MPC` (GPU backend).

    python examples/lidar_path_track_headless.py [world.yaml]
"""
import os
import sys
import time
from collections import namedtuple

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import rda_planner_amd.world as irsim                      # instead of: import irsim
from rda_planner_amd import scenarios as sc
from rda_planner_amd.lidar import scan_box
from rda_planner_amd.mpc import MPC                        # instead of: from RDA_planner.mpc import MPC


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    world = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "world_lidar_track.yaml")
    env = irsim.make(world, save_ani=False, display=False)
    car = namedtuple("car", "G h cone_type wheelbase max_speed max_acce dynamics")
    robot_info = env.get_robot_info()
    car_tuple = car(robot_info.G, robot_info.h, robot_info.cone_type, robot_info.wheelbase, [10, 1], [10, 0.5], "acker")
    ref_path_list = sc.path_track_ref()
    mpc_opt = MPC(car_tuple, ref_path_list, receding=10, sample_time=env.step_time, process_num=4, iter_num=2, max_edge_num=4,
                  max_obs_num=4, obstacle_order=True, wu=1.0, slack_gain=13)
    t0, steps, min_clear, seen = time.perf_counter(), 0, float("inf"), 0
    for i in range(500):
        scan_data = env.get_lidar_scan()
        obs_list = scan_box(env.robot.state, scan_data)
        seen = max(seen, len(obs_list))
        for o in obs_list:
            env.draw_box(o.vertex, refresh=True)
        opt_vel, info = mpc_opt.control(env.robot.state, 4, obs_list)
        env.step(opt_vel)
        env.render(show_traj=True, show_trail=True)
        steps += 1
        min_clear = min(min_clear, env.clearance())
        if env.done():
            break
        if info["arrive"]:
            print("arrive at the goal")
            break
    dt = time.perf_counter() - t0
    print(f"{steps} steps, {steps / dt:.0f} steps/s, up to {seen} boxes per scan, min clearance {min_clear:.2f} m, collided={env.collided}")


if __name__ == "__main__":
    main()
