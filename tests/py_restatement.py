"""A second, independent restatement of the reference's sampler loops in plain Python (small
cases only).  It follows src/resampling.jl / src/solveinitialvalueproblem.jl statement by
statement with 1-based indices, and shares only the RNG contract (uniforms supplied by the
caller) with the C oracle -- so agreement between the two checks the C restatement's indexing,
layout and control flow rather than itself."""
import numpy as np


def _hour(state, trans, p_drive, p_dest, C, Z, t, uniforms, sample_travel=None):
    # t is 1-based like the Julia loop variable
    for i in range(1, C + 1):                                   # resampling.jl:11-22
        origin = state[i - 1, t - 1]
        RndVar = uniforms(i, t)[0]
        driving_probability = p_drive[origin - 1, t - 1]
        if RndVar <= driving_probability:
            drive = 1
        else:
            drive = 0
            trans[i - 1, t - 1, 1] = origin
        trans[i - 1, t - 1, 0] = drive
    for i in range(1, C + 1):                                   # resampling.jl:26-49
        drive = trans[i - 1, t - 1, 0]
        if drive == 1:
            RndVar = uniforms(i, t)[1]
            range_up = 0.0
            range_low = 0.0
            destination = 0
            origin = state[i - 1, t - 1]
            distribution = p_dest[origin - 1, :, t - 1].copy()
            if float(np.sum(distribution)) == 0:
                destination = origin
            else:
                for j in range(1, Z + 1):
                    range_up = range_up + distribution[j - 1]
                    if range_low < RndVar <= range_up:
                        destination = j
                        break
                    range_low = range_up
                if destination == 0:                            # deviation D1 (reference: crash, A-7)
                    nz = np.nonzero(distribution > 0)[0]
                    destination = int(nz[0] + 1) if RndVar == 0.0 else int(nz[-1] + 1)
            trans[i - 1, t - 1, 1] = destination


def solveinitialvalueproblem(state, trans, p_drive, p_dest, C, Z, T, uniforms_ivp):
    for t in range(1, T):                                       # solveinitialvalueproblem.jl:8
        _hour(state, trans, p_drive, p_dest, C, Z, t, uniforms_ivp)
        state[:, t] = np.rint(trans[:, t - 1, 1]).astype(np.int64)   # :53
    return state[:, T - 1].copy()                               # :57-58


def resampling(state, trans, p_drive, p_dest, C, Z, T, uniforms_res):
    for t in range(1, T + 1):                                   # resampling.jl:7
        _hour(state, trans, p_drive, p_dest, C, Z, t, uniforms_res)
        if t < T:                                               # :81-83
            state[:, t] = np.rint(trans[:, t - 1, 1]).astype(np.int64)
    return state, trans


def histogram(Z, T, state, trans, C):
    driving_cars = np.zeros((Z, T))
    parking_cars = np.zeros((Z, T))
    for i in range(C):                                          # saveresults.jl:8-17
        for t in range(T):
            index = state[i, t]
            parking_cars[index - 1, t] += 1
            if trans[i, t, 0] == 1:
                driving_cars[index - 1, t] += 1
    return parking_cars, driving_cars
