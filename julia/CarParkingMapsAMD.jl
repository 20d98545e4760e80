# CarParkingMapsAMD.jl -- ccall shim over libcpm_hip.so (C ABI: include/cpm.h).
#
# Drop-in for the sampler path of main.jl:82-102: defines createpdrive, createpdestin,
# initializestates, solveinitialvalueproblem, resampling, averagedrivingtime and saveresults with
# the reference's positional signatures, so `include("julia/CarParkingMapsAMD.jl")` placed AFTER
# main.jl's own includes (main.jl:11-22) overrides them and the rest of main.jl runs unchanged.
# Like the reference's functions it reads the script globals T, cars_per_zone, p_min, p_max,
# e_drive, e_dest (main.jl:37-42); two optional extra globals: CPM_SEED (UInt64) and CPM_DEVICE.
#
# NOT EXECUTED IN THE BUILD IMAGE: Julia is absent there and on the GPU box.  The identical C
# symbols are exercised through ctypes by tests/ (Python host mirror, carparkingmaps_amd/).
# No CUDA.jl, no AMDGPU.jl: plain ccall.

module CarParkingMapsAMD

const libcpm = get(ENV, "CPM_LIB", joinpath(@__DIR__, "..", "carparkingmaps_amd", "csrc", "libcpm_hip.so"))

mutable struct Ctx
    h::Ptr{Cvoid}
    Z::Int
    T::Int
end

const _ctx = Dict{Tuple{Int,Int,Int},Ctx}()
const _last = Dict{UInt,Any}()          # objectid(state_matrix) => (parking, driving, sum_tt_q16, C)

function _check(status::Cint)
    if status != 0
        msg = unsafe_string(ccall((:cpm_last_error, libcpm), Cstring, ()))
        error("libcpm_hip status $status: $msg")
    end
    nothing
end

function context(Z::Integer, T::Integer, device::Integer=0)
    get!(_ctx, (Int(Z), Int(T), Int(device))) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall((:cpm_create, libcpm), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Cint), h, Z, T, device))
        Ctx(h[], Int(Z), Int(T))
    end
end

function release()
    for c in values(_ctx)
        ccall((:cpm_destroy, libcpm), Cint, (Ptr{Cvoid},), c.h)
    end
    empty!(_ctx); empty!(_last)
end

_seed() = isdefined(Main, :CPM_SEED) ? UInt64(Main.CPM_SEED) : UInt64(0x5EEDCA125)
_dev() = isdefined(Main, :CPM_DEVICE) ? Int(Main.CPM_DEVICE) : 0

# ---- device-resident datamatrix / distance matrix (src/createdatamatrix.jl:3-27, src/processgeodata.jl:148-166) ----
# createdatamatrix() below reads the CSV natively and builds the Z x Z x T x 2 array in HBM; what it returns stands for
# that array in the calls main.jl makes with it (createpdrive, createpdestin, resampling).
struct DeviceArray
    kind::Symbol
    number_zones::Int
end

function createdatamatrix(path_to_csv_data, number_zones)
    c = context(number_zones, Main.T, _dev())
    n = Ref{Int64}(0)
    _check(ccall((:cpm_createdatamatrix_csv, libcpm), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), c.h, path_to_csv_data, n))
    DeviceArray(:datamatrix, Int(number_zones))
end

# the distance part of processgeodata on the device, from the centroids the reference computes (:99-146)
function distance_from_centroids(centroid_lat::Vector{Float64}, centroid_long::Vector{Float64}, number_zones)
    c = context(number_zones, Main.T, _dev())
    GC.@preserve centroid_lat centroid_long begin
        _check(ccall((:cpm_set_distance_from_centroids, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, centroid_lat, centroid_long))
    end
    DeviceArray(:distance, Int(number_zones))
end

function _use(c::Ctx, datamatrix, distance_matrix_km)
    if datamatrix isa DeviceArray
        if distance_matrix_km isa Matrix{Float64}      # the reference's processgeodata result beside the device datamatrix
            GC.@preserve distance_matrix_km _check(ccall((:cpm_set_distance, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, distance_matrix_km))
        end
    else
        GC.@preserve datamatrix distance_matrix_km begin
            _check(ccall((:cpm_set_datamatrix, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, datamatrix, distance_matrix_km))
        end
    end
end

# src/createpdrive.jl:3-38
function createpdrive(datamatrix::DeviceArray, distance_matrix_km, number_zones)
    c = context(number_zones, Main.T, _dev())
    _use(c, datamatrix, distance_matrix_km)
    p_drive = zeros(Float64, c.Z, c.T)
    _check(ccall((:cpm_build_p_drive, libcpm), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
                 c.h, Main.p_min, Main.p_max, Main.e_drive, p_drive))
    p_drive
end

function createpdrive(datamatrix::Array{Float64,4}, distance_matrix_km::Matrix{Float64}, number_zones)
    c = context(number_zones, Main.T, _dev())
    GC.@preserve datamatrix distance_matrix_km begin
        _check(ccall((:cpm_set_datamatrix, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                     c.h, datamatrix, distance_matrix_km))
    end
    p_drive = zeros(Float64, c.Z, c.T)
    _check(ccall((:cpm_build_p_drive, libcpm), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
                 c.h, Main.p_min, Main.p_max, Main.e_drive, p_drive))
    p_drive
end

# src/createpdestin.jl:3-50 (datamatrix was uploaded by createpdrive, main.jl:82 runs first)
function createpdestin(datamatrix::Union{Array{Float64,4},DeviceArray}, number_zones)
    c = context(number_zones, Main.T, _dev())
    p_dest = zeros(Float64, c.Z, c.Z, c.T)
    _check(ccall((:cpm_build_p_dest, libcpm), Cint, (Ptr{Cvoid}, Float64, Cint, Ptr{Float64}),
                 c.h, Float64(Main.e_dest), Main.e_dest isa Integer ? 1 : 0, p_dest))
    p_dest
end

# src/initializestates.jl:4-22 -- host arrays exactly as the reference allocates them
function initializestates(C)
    T = Main.T; cpz = Main.cars_per_zone
    state_matrix = zeros(Int64, C, T)
    transition_matrix = zeros(Float64, C, T, 4)
    zone = 0
    for i = 0:cpz:(C - cpz)
        zone += 1
        state_matrix[i+1:i+cpz, 1] .= zone
    end
    state_matrix, transition_matrix
end

function _install(c::Ctx, p_drive, p_dest)
    GC.@preserve p_drive p_dest begin
        _check(ccall((:cpm_set_p_drive, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, p_drive))
        _check(ccall((:cpm_set_p_dest, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, p_dest))
    end
end

# src/solveinitialvalueproblem.jl:4-62
function solveinitialvalueproblem(state_matrix, transition_matrix, p_drive, p_dest, C, number_zones)
    c = context(number_zones, Main.T, _dev())
    _install(c, p_drive, p_dest)
    _check(ccall((:cpm_init_states, libcpm), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64), c.h, C, Main.cars_per_zone, 0, C))
    zones = state_matrix[:, 1]
    _check(ccall((:cpm_set_state, libcpm), Cint, (Ptr{Cvoid}, Ptr{Int64}), c.h, zones))
    initial_state = zeros(Int64, C)
    _check(ccall((:cpm_solve_ivp, libcpm), Cint, (Ptr{Cvoid}, UInt64, Ptr{Int64}), c.h, _seed(), initial_state))
    initial_state
end

# src/resampling.jl:3-89 -- fills both matrices in place and returns them
function resampling(state_matrix, transition_matrix, C, number_zones, p_drive, p_dest, datamatrix, distance_matrix_km)
    c = context(number_zones, Main.T, _dev())
    _install(c, p_drive, p_dest)
    _use(c, datamatrix, distance_matrix_km)
    _check(ccall((:cpm_init_states, libcpm), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64), c.h, C, Main.cars_per_zone, 0, C))
    zones = state_matrix[:, 1]
    _check(ccall((:cpm_set_state, libcpm), Cint, (Ptr{Cvoid}, Ptr{Int64}), c.h, zones))
    parking = zeros(Int64, c.Z, c.T); driving = zeros(Int64, c.Z, c.T); tt = Ref{Int64}(0)
    GC.@preserve state_matrix transition_matrix begin
        _check(ccall((:cpm_resample, libcpm), Cint,
                     (Ptr{Cvoid}, UInt64, UInt32, Ptr{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}, Ptr{Float64}),
                     c.h, _seed(), UInt32(1), parking, driving, tt, state_matrix, transition_matrix))
    end
    _last[objectid(state_matrix)] = (parking, driving, tt[], C)
    state_matrix, transition_matrix
end

# src/averagedrivingtime.jl:3-12
function averagedrivingtime(C, A_drive, transition_matrix)
    T = Main.T
    A_drive + sum(@view transition_matrix[:, :, 3]) / (C * T * 60 * 60)
end

# src/saveresults.jl:6-28 -- the zone x hour histogram comes from the fused device result
function zone_hour_counts(state_matrix)
    haskey(_last, objectid(state_matrix)) || error("saveresults: matrices were not produced by the last resampling() call")
    _last[objectid(state_matrix)]
end

end # module

# Override the reference's definitions (include this file after main.jl:11-22).
createpdrive(dm, dist, Z) = CarParkingMapsAMD.createpdrive(dm, dist, Z)
createpdestin(dm, Z) = CarParkingMapsAMD.createpdestin(dm, Z)
initializestates(C) = CarParkingMapsAMD.initializestates(C)
solveinitialvalueproblem(s, tr, pd, pde, C, Z) = CarParkingMapsAMD.solveinitialvalueproblem(s, tr, pd, pde, C, Z)
resampling(s, tr, C, Z, pd, pde, dm, dist) = CarParkingMapsAMD.resampling(s, tr, C, Z, pd, pde, dm, dist)
averagedrivingtime(C, A, tr) = CarParkingMapsAMD.averagedrivingtime(C, A, tr)
# Optional (uncomment to keep the 6.4 GB datamatrix off the host): the CSV is then read by the library's own reader.
# createdatamatrix(path_to_csv_data, number_zones) = CarParkingMapsAMD.createdatamatrix(path_to_csv_data, number_zones)

# saveresults keeps the reference's CSV tail (src/saveresults.jl:23-42) and takes the counts from the device.
function saveresults(number_zones, state_matrix, transition_matrix, path_to_results, data_set, C)
    parking, driving, _, _ = CarParkingMapsAMD.zone_hour_counts(state_matrix)
    parking_cars = parking ./ C                                            # src/saveresults.jl:20
    traffic_resultsmatrix = Float64.(sum(driving, dims=1))                 # :23
    min_sampled = minimum(traffic_resultsmatrix); max_sampled = maximum(traffic_resultsmatrix)
    for i = 1:24
        traffic_resultsmatrix[i] = (traffic_resultsmatrix[i] - min_sampled) / (max_sampled - min_sampled)
    end
    header_vector = [string("t = ", t, "h") for t = 1:T]
    CSV.write(string(path_to_results, "/results_parkingdensities_", data_set), DataFrame(parking_cars, :auto), header=header_vector)
    CSV.write(string(path_to_results, "/results_trafficactivity_", data_set), DataFrame(traffic_resultsmatrix, :auto), header=header_vector)
end
