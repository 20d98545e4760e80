# CarParkingMapsAMD.jl -- ccall shim over libcpm_hip.so (C ABI: include/cpm.h).
#
# Drop-in for the sampler path of main.jl:82-95: defines createpdrive, createpdestin, solveinitialvalueproblem and resampling
# with the reference's positional signatures, so `include("julia/CarParkingMapsAMD.jl")` placed AFTER main.jl's own includes
# (main.jl:11-22) overrides those four and the rest of main.jl runs unchanged -- the reference's own initializestates,
# averagedrivingtime and saveresults keep working on the matrices `resampling` fills (compat mode).  Like the reference's
# functions it reads the script globals T, cars_per_zone, p_min, p_max, e_drive, e_dest (main.jl:37-42) from Main at call time;
# optional extra globals: CPM_SEED (UInt64), CPM_DEVICE, CPM_TRUST_UNCHANGED (skip the content check of host arrays, below).
#
# Fast mode (no C x T matrices on the host): `CarParkingMapsAMD.run_dataset(datamatrix, distance_matrix_km, number_zones)`.
#
# NOT EXECUTED IN THE BUILD IMAGE: Julia is absent there and on the GPU box.  The identical C symbols, argument order and array
# layouts are exercised through ctypes (carparkingmaps_amd/) and from plain C (tests/abi_harness.c) by tests/.
# No CUDA.jl, no AMDGPU.jl: plain ccall.

module CarParkingMapsAMD

const libcpm = get(ENV, "CPM_LIB", joinpath(@__DIR__, "..", "carparkingmaps_amd", "csrc", "libcpm_hip.so"))

mutable struct Ctx
    h::Ptr{Cvoid}
    Z::Int
    T::Int
    resident::Dict{Symbol,Tuple{UInt,Tuple,UInt64,UInt64}}   # table name => (address, size, content fingerprint) of the host array last uploaded
end

const _ctx = Dict{Tuple{Int,Int,Int},Ctx}()

function _check(status::Cint)
    if status != 0
        msg = unsafe_string(ccall((:cpm_last_error, libcpm), Cstring, ()))
        error("libcpm_hip status $status: $msg")
    end
    nothing
end

_T() = Int(Main.T)
_seed() = isdefined(Main, :CPM_SEED) ? UInt64(Main.CPM_SEED) : UInt64(0x5EEDCA125)
_dev() = isdefined(Main, :CPM_DEVICE) ? Int(Main.CPM_DEVICE) : 0
_trust() = isdefined(Main, :CPM_TRUST_UNCHANGED) && Main.CPM_TRUST_UNCHANGED === true

function context(Z::Integer, T::Integer=_T(), device::Integer=_dev())
    get!(_ctx, (Int(Z), Int(T), Int(device))) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall((:cpm_create, libcpm), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Cint), h, Z, T, device))
        Ctx(h[], Int(Z), Int(T), Dict{Symbol,Tuple{UInt,Tuple,UInt64,UInt64}}())
    end
end

function release()
    for c in values(_ctx)
        ccall((:cpm_destroy, libcpm), Cint, (Ptr{Cvoid},), c.h)
    end
    empty!(_ctx)
end

# Forget what is resident: the next call uploads its host arrays whatever they hold.
invalidate() = foreach(c -> empty!(c.resident), values(_ctx))

# Is `a` (a host array) what the device already holds under `name`?  Same address and size, and -- unless CPM_TRUST_UNCHANGED --
# the same content: one pass over the array that depends on WHERE every word sits (a wrapping UInt64 sum of the words and a
# wrapping sum of word x (2 x position + 1): an edit in place changes it, and so does a permutation in place -- two zones' rows
# swapped, a sort --, which a plain sum would not see).  Far below the cost of the PCIe upload + table build it saves.
function _fingerprint(a::Array{Float64})
    w = reinterpret(UInt64, vec(a))
    plain = UInt64(0); weighted = UInt64(0)
    @inbounds for i in eachindex(w)
        plain += w[i]
        weighted += w[i] * (UInt64(2) * UInt64(i - 1) + UInt64(1))
    end
    (plain, weighted)
end
_stamp(a::Array{Float64}) = (UInt(pointer(a)), size(a), (_trust() ? (UInt64(0), UInt64(0)) : _fingerprint(a))...)
function _ensure(upload::Function, c::Ctx, name::Symbol, a::Array{Float64})
    s = _stamp(a)
    if get(c.resident, name, nothing) != s
        upload()
        c.resident[name] = s
    end
    nothing
end

# ---- device-resident datamatrix / distance matrix (src/createdatamatrix.jl:3-27, src/processgeodata.jl:148-166) ----
# createdatamatrix() below reads the CSV natively and builds the Z x Z x T x 2 array in HBM; what it returns stands for
# that array in the calls main.jl makes with it (createpdrive, createpdestin, resampling).
struct DeviceArray
    kind::Symbol
    number_zones::Int
end

function createdatamatrix(path_to_csv_data, number_zones)
    c = context(number_zones)
    n = Ref{Int64}(0)
    _check(ccall((:cpm_createdatamatrix_csv, libcpm), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), c.h, path_to_csv_data, n))
    delete!(c.resident, :datamatrix)
    DeviceArray(:datamatrix, Int(number_zones))
end

# the distance part of processgeodata on the device, from the centroids the reference computes (:99-146)
function distance_from_centroids(centroid_lat::Vector{Float64}, centroid_long::Vector{Float64}, number_zones)
    c = context(number_zones)
    GC.@preserve centroid_lat centroid_long begin
        _check(ccall((:cpm_set_distance_from_centroids, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, centroid_lat, centroid_long))
    end
    delete!(c.resident, :distance)
    DeviceArray(:distance, Int(number_zones))
end

# make `datamatrix` (and the distance matrix, when one is given) the arrays resident in the context; host arrays that are
# already resident and unchanged are not uploaded again (6.4 GB at Z = 4,096)
function _use(c::Ctx, datamatrix, distance_matrix_km)
    if distance_matrix_km isa Matrix{Float64}
        _ensure(c, :distance, distance_matrix_km) do
            GC.@preserve distance_matrix_km _check(ccall((:cpm_set_distance, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, distance_matrix_km))
        end
    end
    if datamatrix isa Array{Float64,4}
        size(datamatrix) == (c.Z, c.Z, c.T, 2) || error("datamatrix: expected $((c.Z, c.Z, c.T, 2)), got $(size(datamatrix))")
        _ensure(c, :datamatrix, datamatrix) do       # (NULL distance matrix: the resident one is kept)
            GC.@preserve datamatrix _check(ccall((:cpm_set_datamatrix, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, datamatrix, C_NULL))
        end
    elseif !(datamatrix isa DeviceArray)
        error("datamatrix: expected Array{Float64,4} or the DeviceArray createdatamatrix returned")
    end
    nothing
end

# src/createpdrive.jl:3-38
function createpdrive(datamatrix, distance_matrix_km, number_zones)
    c = context(number_zones)
    _use(c, datamatrix, distance_matrix_km)
    p_drive = zeros(Float64, c.Z, c.T)
    _check(ccall((:cpm_build_p_drive, libcpm), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
                 c.h, Float64(Main.p_min), Float64(Main.p_max), Float64(Main.e_drive), p_drive))
    c.resident[:p_drive] = _stamp(p_drive)      # what the device holds IS this array
    p_drive
end

# src/createpdestin.jl:3-50 -- builds from the datamatrix it is GIVEN (uploaded now unless it is the resident one, unchanged)
function createpdestin(datamatrix, number_zones)
    c = context(number_zones)
    _use(c, datamatrix, nothing)
    p_dest = zeros(Float64, c.Z, c.Z, c.T)
    _check(ccall((:cpm_build_p_dest, libcpm), Cint, (Ptr{Cvoid}, Float64, Cint, Ptr{Float64}),
                 c.h, Float64(Main.e_dest), Main.e_dest isa Integer ? 1 : 0, p_dest))
    c.resident[:p_dest] = _stamp(p_dest)
    p_dest
end

function _install(c::Ctx, p_drive::Matrix{Float64}, p_dest::Array{Float64,3})
    _ensure(c, :p_drive, p_drive) do
        GC.@preserve p_drive _check(ccall((:cpm_set_p_drive, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, p_drive))
    end
    _ensure(c, :p_dest, p_dest) do
        GC.@preserve p_dest _check(ccall((:cpm_set_p_dest, libcpm), Cint, (Ptr{Cvoid}, Ptr{Float64}), c.h, p_dest))
    end
end

function _place_cars(c::Ctx, state_matrix, C)
    _check(ccall((:cpm_init_states, libcpm), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64), c.h, C, Int64(Main.cars_per_zone), 0, C))
    zones = Vector{Int64}(state_matrix[:, 1])           # state_matrix[:,1], 1-based zone ids
    GC.@preserve zones _check(ccall((:cpm_set_state, libcpm), Cint, (Ptr{Cvoid}, Ptr{Int64}), c.h, zones))
end

# src/solveinitialvalueproblem.jl:4-62
function solveinitialvalueproblem(state_matrix, transition_matrix, p_drive, p_dest, C, number_zones)
    c = context(number_zones)
    _install(c, p_drive, p_dest)
    _place_cars(c, state_matrix, C)
    initial_state = zeros(Int64, C)
    _check(ccall((:cpm_solve_ivp, libcpm), Cint, (Ptr{Cvoid}, UInt64, Ptr{Int64}), c.h, _seed(), initial_state))
    initial_state
end

# src/resampling.jl:3-89 -- fills both matrices in place and returns them (compat mode: the reference's saveresults and
# averagedrivingtime then work on them unchanged)
function resampling(state_matrix::Matrix{Int64}, transition_matrix::Array{Float64,3}, C, number_zones, p_drive, p_dest, datamatrix, distance_matrix_km)
    c = context(number_zones)
    _install(c, p_drive, p_dest)
    _use(c, datamatrix, distance_matrix_km)
    _place_cars(c, state_matrix, C)
    parking = zeros(Int64, c.Z, c.T); driving = zeros(Int64, c.Z, c.T); tt = Ref{Int64}(0)
    GC.@preserve state_matrix transition_matrix begin
        _check(ccall((:cpm_resample, libcpm), Cint,
                     (Ptr{Cvoid}, UInt64, UInt32, Ptr{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}, Ptr{Float64}),
                     c.h, _seed(), UInt32(1), parking, driving, tt, state_matrix, transition_matrix))
    end
    state_matrix, transition_matrix
end

# ---- fast mode: main.jl:82-98 for one dataset without the C x T matrices (0.8 GB + 3.1 GB at Z = 4,096) ever leaving the device ----
# Returns (parking_cars = counts ./ C (src/saveresults.jl:20), traffic activity (the min-max normalised column sums, :23-28),
# the increment of A_drive (src/averagedrivingtime.jl:10)).  The caller writes them with its own CSV lines (src/saveresults.jl:31-42).
function run_dataset(datamatrix, distance_matrix_km, number_zones)
    c = context(number_zones)
    _use(c, datamatrix, distance_matrix_km)
    _check(ccall((:cpm_build_p_drive, libcpm), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
                 c.h, Float64(Main.p_min), Float64(Main.p_max), Float64(Main.e_drive), C_NULL))
    _check(ccall((:cpm_build_p_dest, libcpm), Cint, (Ptr{Cvoid}, Float64, Cint, Ptr{Float64}),
                 c.h, Float64(Main.e_dest), Main.e_dest isa Integer ? 1 : 0, C_NULL))
    delete!(c.resident, :p_drive); delete!(c.resident, :p_dest)
    C = Int64(number_zones) * Int64(Main.cars_per_zone)
    _check(ccall((:cpm_init_states, libcpm), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64), c.h, C, Int64(Main.cars_per_zone), 0, C))
    _check(ccall((:cpm_solve_ivp, libcpm), Cint, (Ptr{Cvoid}, UInt64, Ptr{Int64}), c.h, _seed(), C_NULL))
    parking = zeros(Int64, c.Z, c.T); driving = zeros(Int64, c.Z, c.T); tt = Ref{Int64}(0)
    _check(ccall((:cpm_resample, libcpm), Cint,
                 (Ptr{Cvoid}, UInt64, UInt32, Ptr{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}, Ptr{Float64}),
                 c.h, _seed(), UInt32(1), parking, driving, tt, C_NULL, C_NULL))
    activity = vec(sum(driving, dims=1)) .* 1.0
    lo, hi = extrema(activity)
    (parking ./ C, (activity .- lo) ./ (hi - lo), (tt[] / 65536) / (C * c.T * 3600))
end

end # module

# Override the reference's definitions (include this file after main.jl:11-22).  initializestates, averagedrivingtime and
# saveresults stay the reference's own.
createpdrive(dm, dist, Z) = CarParkingMapsAMD.createpdrive(dm, dist, Z)
createpdestin(dm, Z) = CarParkingMapsAMD.createpdestin(dm, Z)
solveinitialvalueproblem(s, tr, pd, pde, C, Z) = CarParkingMapsAMD.solveinitialvalueproblem(s, tr, pd, pde, C, Z)
resampling(s, tr, C, Z, pd, pde, dm, dist) = CarParkingMapsAMD.resampling(s, tr, C, Z, pd, pde, dm, dist)
# Optional (uncomment to keep the 6.4 GB datamatrix off the host): the CSV is then read by the library's own reader.
# createdatamatrix(path_to_csv_data, number_zones) = CarParkingMapsAMD.createdatamatrix(path_to_csv_data, number_zones)
