#!/usr/bin/env python3
"""main.jl of the reference, line for line, on the Python host mirror (carparkingmaps_amd): for every city directory
under `path_to_cities` (N Uber Movement CSV files + one GeoJSON that sorts last, main.jl:51-53) sample every dataset and write
results_parkingdensities_<csv>, results_trafficactivity_<csv>, zoneID_coordinates.csv and sampling_parameters.csv into
`path_to_results_folder/<city>/`.  The dense datamatrix / distance matrix / p_dest are built and kept in HBM.

    python examples/main.py /data/uber/ /data/results/ [--cars-per-zone 1000] [--seed 0x5EEDCA125] [--compat]

--compat materialises state_matrix / transition_matrix on the host exactly as main.jl:88-102 does (C x T x 5 values: only for
small fleets); without it the zone x hour counts and the travel-time sum come straight from the device (same numbers).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import reference_api as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path_to_cities")
    ap.add_argument("path_to_results_folder")
    ap.add_argument("--cars-per-zone", type=int, default=1000)   # main.jl:41
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EEDCA125)
    ap.add_argument("--compat", action="store_true")
    args = ap.parse_args()
    path_to_cities = os.path.join(args.path_to_cities, "")
    path_to_results_folder = os.path.join(args.path_to_results_folder, "")
    # main.jl:37-42
    R.params.e_drive, R.params.e_dest, R.params.p_min, R.params.p_max = 0.5, 2, 0.1, 0.9
    R.params.cars_per_zone, R.params.T, R.params.seed = args.cars_per_zone, 24, args.seed
    T = R.params.T
    for city in sorted(os.listdir(path_to_cities)):                                   # main.jl:45
        path_to_data = path_to_cities + city
        if not os.path.isdir(path_to_data):
            continue
        dataset_list = sorted(os.listdir(path_to_data))                               # main.jl:51
        path_to_json_data = os.path.join(path_to_data, dataset_list[-1])              # main.jl:52
        csv_dataset_list = dataset_list[:-1]                                          # main.jl:53
        path_to_results = R.createresultsdirectory(path_to_results_folder, city)      # main.jl:56
        distance_matrix_km, number_zones = R.processgeodata(path_to_json_data, path_to_data, csv_dataset_list, path_to_results)  # :59
        C = int(number_zones * R.params.cars_per_zone)                                # main.jl:62-63
        print(f"Simulation starts for the city of {city} with a vehicle fleet of C = {C}")
        A_drive, count_dataset = 0.0, 0                                               # main.jl:67-68
        for data_set in csv_dataset_list:                                             # main.jl:71
            path_to_csv_data = os.path.join(path_to_data, data_set)
            print("Simulating", data_set)
            datamatrix = R.createdatamatrix(path_to_csv_data, number_zones)           # main.jl:79
            if args.compat:
                p_drive = R.createpdrive(datamatrix, distance_matrix_km, number_zones)             # main.jl:82
                p_dest = R.createpdestin(datamatrix, number_zones)                                 # main.jl:85
                state_matrix, transition_matrix = R.initializestates(C)                            # main.jl:88
                state_matrix[:, 0] = R.solveinitialvalueproblem(state_matrix, transition_matrix, p_drive, p_dest, C, number_zones)  # :91-92
                state_matrix, transition_matrix = R.resampling(state_matrix, transition_matrix, C, number_zones, p_drive, p_dest,
                                                               datamatrix, distance_matrix_km)     # main.jl:95
                A_drive = R.averagedrivingtime(C, A_drive, transition_matrix)                      # main.jl:98
                R.saveresults(number_zones, state_matrix, transition_matrix, path_to_results, data_set, C)  # main.jl:102
            else:
                out = R.run_dataset(datamatrix, distance_matrix_km, number_zones, travel=True)     # main.jl:82-98 on the device
                A_drive += out["A_drive_increment"]
                header = ",".join(f"t = {t}h" for t in range(1, T + 1))                            # src/saveresults.jl:34-38
                with open(os.path.join(path_to_results, "results_parkingdensities_" + data_set), "w") as f:
                    f.write(header + "\n")
                    for z in range(number_zones):
                        f.write(",".join(R.julia_float(v) for v in out["parking_density"][z]) + "\n")
                with open(os.path.join(path_to_results, "results_trafficactivity_" + data_set), "w") as f:
                    f.write(header + "\n" + ",".join(R.julia_float(v) for v in out["traffic_activity"]) + "\n")
            count_dataset += 1                                                        # main.jl:99
        A_drive = A_drive / count_dataset if count_dataset else float("nan")         # main.jl:107
        R.saveparameters(path_to_results, T, number_zones, R.params.cars_per_zone, C, R.params.e_drive, R.params.p_min, R.params.p_max,
                         R.params.e_dest, A_drive)                                    # main.jl:110
        R.release()


if __name__ == "__main__":
    main()
