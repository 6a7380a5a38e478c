# CameraCalibrationsAMD.jl -- thin `ccall` binding of libccal_hip.so (include/ccal.h).
#
# UNTESTED IN THIS REPOSITORY: the build image has no Julia -- none of this file has ever been parsed or run.  The Python mirror
# (cameracalibrations_amd/solver.py, tested on MI355X) makes exactly the same calls with the
# same array layouts; keep the two in step.  What IS checked without Julia (tests/test_cabi.py): every `ccall` in this
# file names a symbol of include/ccal.h and passes as many arguments, of the same C width, as cameracalibrations_amd/_ffi.py
# declares for it.  Everything below `fit_model` -- `Context`, the map calls, `Calibration`, `fit`, `fit_corners` -- is
# EXPERIMENTAL surface on top of the seam (INTEGRATION.md).
#
# Drop-in for the reference's FFI seam: define `fit_model` with the signature of
# reference src/detect_fit.jl:27 and return the NamedTuple of :60.  In CameraCalibrations.jl
# replace the body of `fit_model` by `CameraCalibrationsAMD.fit_model(args...)`, or
#     @eval CameraCalibrations fit_model(args...) = CameraCalibrationsAMD.fit_model(args...)
module CameraCalibrationsAMD

using StaticArrays

const LIB = get(ENV, "CCAL_LIB", joinpath(@__DIR__, "..", "cameracalibrations_amd", "libccal_hip.so"))

# OpenCV flag bits used by the reference (src/detect_fit.jl:40)
const ZERO_TANGENT_DIST = 0x00000008
const FIX_ASPECT_RATIO  = 0x00000002
const FIX_K1 = 0x00000020
const FIX_K2 = 0x00000040
const FIX_K3 = 0x00000080
const CRITERIA = (30, 0.001)          # src/CameraCalibrations.jl:16

struct CcalError <: Exception
    code::Int32
    msg::String
end

function check(rc::Int32)
    rc < 0 && throw(CcalError(rc, unsafe_string(ccall((:ccal_last_error, LIB), Cstring, ()))))
    return rc
end

"""
    fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect)

Same contract as the reference wrapper of `OpenCV.calibrateCamera`: `objpoints` is the
n1 x n2 x 1 template of XYZ, `imgpointss` a vector of n1 x n2 matrices of RowCol, `sz` the
size of the Julia image (== OpenCV (width, height), src/detect_fit.jl:32).
"""
function fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect; device = 0, devices = nothing)
    P = prod(n_corners)
    N = length(imgpointss)
    obj = Float32.(reshape(reinterpret(reshape, eltype(eltype(objpoints)), vec(objpoints)), 3, P))   # [3,P]
    img = Array{Float32}(undef, 2, P, N)                                                         # [2,P,N]
    for (i, m) in enumerate(imgpointss), (p, rc) in enumerate(vec(m))
        img[1, p, i] = rc[1]; img[2, p, i] = rc[2]
    end
    flags = ZERO_TANGENT_DIST | FIX_K3 | FIX_K2 | (with_distortion ? 0x0 : FIX_K1) | FIX_ASPECT_RATIO
    K4 = zeros(Float64, 4); dist = zeros(Float64, 5)
    rvecs = zeros(Float64, 3, N); tvecs = zeros(Float64, 3, N)
    rms = Ref{Float64}(0); iters = Ref{Int32}(0)
    if devices !== nothing      # several GPUs of the node from this one task: images sharded over `devices`
        devs = Int32.(collect(devices))
        check(ccall((:ccal_calibrate_multi, LIB), Int32,
                    (Ptr{Float32}, Ptr{Float32}, Int32, Int32, Int32, Int32, Float64, UInt32, Int32, Float64, Int32, Ptr{Int32},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int32}, Ptr{Cvoid}),
                    obj, img, P, N, sz[1], sz[2], Float64(aspect), flags, CRITERIA[1], CRITERIA[2], length(devs), devs,
                    K4, dist, rvecs, tvecs, rms, iters, C_NULL))
    else
    check(ccall((:ccal_calibrate, LIB), Int32,
                (Ptr{Float32}, Ptr{Float32}, Int32, Int32, Int32, Int32, Float64, UInt32, Int32, Float64, Int32,
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int32}, Ptr{Cvoid}),
                obj, img, P, N, sz[1], sz[2], Float64(aspect), flags, CRITERIA[1], CRITERIA[2], device,
                K4, dist, rvecs, tvecs, rms, iters, C_NULL))
    end
    k = dist[1]
    Rs = [rvecs[:, i] for i in 1:N]
    ts = [tvecs[:, i] for i in 1:N]
    frow, fcol, crow, ccol = K4
    return (; k, Rs, ts, frow, fcol, crow, ccol)
end

# Batched point maps (src/meta.jl:82,88): xyz [3,M] <-> rowcol [2,M] for one image's extrinsics.
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ccal_create, LIB), Int32, (Int32, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[])
        finalizer(x -> ccall((:ccal_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), c)
        return c
    end
end

"""
    fit_models(sz, objpoints, imgpointsss, n_corners, with_distortion, aspect)

`fit_model` for several calibrations at once (one vector of per-image corner matrices per calibration; template, image size and
flags shared) in one call of `ccal_calibrate_many` -- the loop a caller with many cameras runs around
src/buildcalibrations.jl:12-26; small fits share merged launches on the device.  Returns a vector of `fit_model`'s NamedTuples.
"""
function fit_models(sz, objpoints, imgpointsss, n_corners, with_distortion, aspect; device = 0)
    P = prod(n_corners)
    obj = Float32.(reshape(reinterpret(reshape, eltype(eltype(objpoints)), vec(objpoints)), 3, P))   # [3,P]
    flags = ZERO_TANGENT_DIST | FIX_K3 | FIX_K2 | (with_distortion ? 0x0 : FIX_K1) | FIX_ASPECT_RATIO
    imgs = map(imgpointsss) do imgpointss
        img = Array{Float32}(undef, 2, P, length(imgpointss))                                    # [2,P,N]
        for (i, m) in enumerate(imgpointss), (p, rc) in enumerate(vec(m))
            img[1, p, i] = rc[1]; img[2, p, i] = rc[2]
        end
        img
    end
    Ns = Int32[size(img, 3) for img in imgs]
    outs = [zeros(Float64, 11 + 6 * N) for N in Ns]           # K4 | dist5 | rms | iters | rvecs [3,N] | tvecs [3,N]
    GC.@preserve imgs outs begin
        ip = Ptr{Cvoid}[pointer(img) for img in imgs]
        op = Ptr{Cvoid}[pointer(o) for o in outs]
        check(ccall((:ccal_calibrate_many, LIB), Int32,
                    (Ptr{Float32}, Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Int32}, Int32, Int32, Float64, UInt32, Int32, Float64, Int32,
                     Ptr{Ptr{Cvoid}}, Ptr{Int32}),
                    obj, P, length(imgs), ip, Ns, sz[1], sz[2], Float64(aspect), flags, CRITERIA[1], CRITERIA[2], device, op, C_NULL))
    end
    return map(zip(outs, Ns)) do (o, N)
        frow, fcol, crow, ccol = o[1:4]
        k = o[5]
        rv = reshape(o[12:11 + 3N], 3, N); tv = reshape(o[12 + 3N:end], 3, N)
        (; k, Rs = [rv[:, i] for i in 1:N], ts = [tv[:, i] for i in 1:N], frow, fcol, crow, ccol)
    end
end

function project(c::Context, xyz::Matrix{Float64}, f, cc, k, rvec, tvec, checker_size)
    M = size(xyz, 2); out = zeros(Float64, 2, M)
    check(ccall((:ccal_project, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                c.h, xyz, M, Float64[f...], Float64[cc...], Float64(k), Float64[rvec...], Float64[tvec...], Float64(checker_size), out))
    return out
end

function unproject(c::Context, rowcol::Matrix{Float64}, f, cc, k, rvec, tvec, checker_size)
    M = size(rowcol, 2); out = zeros(Float64, 3, M)
    check(ccall((:ccal_unproject, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                c.h, rowcol, M, Float64[f...], Float64[cc...], Float64(k), Float64[rvec...], Float64[tvec...], Float64(checker_size), out))
    return out
end

# calculate_errors (src/buildcalibrations.jl:37-67) on the device.  objpoints [3,P] already scaled by
# checker_size (:15), imgpoints [2,P,N] Float64, inv_pts [2,S,N] round-trip samples (the reference draws
# them from the global RNG, :54; here the caller supplies them so the metric is reproducible).
function calculate_errors(c::Context, n_corners, objpoints::Matrix{Float64}, imgpoints::Array{Float64,3}, f, cc, k,
                          rvecs::Matrix{Float64}, tvecs::Matrix{Float64}, checker_size, inv_pts::Array{Float64,3})
    N = size(imgpoints, 3); S = size(inv_pts, 2); out = zeros(Float64, 4)
    check(ccall((:ccal_errors, LIB), Int32,
                (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64,
                 Ptr{Float64}, Ptr{Float64}, Float64, Int32, Ptr{Float64}, Ptr{Float64}),
                c.h, N, n_corners[1], n_corners[2], objpoints, imgpoints, Float64[f...], Float64[cc...], Float64(k),
                rvecs, tvecs, Float64(checker_size), S, inv_pts, out))
    return (; n = N, reprojection = out[1], projection = out[2], distance = out[3], inverse = out[4])
end

# _detect_corners (src/detect_fit.jl:5-21) on the device: gray is the column-major Matrix{UInt8} of the image
# (element (row, col) at gray[row + rows*col], i.e. stride_row = 1, stride_col = rows -- what the reference
# hands to OpenCV, :7).  Returns `missing` or the n1 x n2 matrix of (row, col) corners, 0-based pixel centres.
function detect_corners(c::Context, gray::Matrix{UInt8}, n_corners)
    rows, cols = size(gray); n1, n2 = n_corners
    corners = zeros(Float64, 2, n1 * n2); found = Ref{Int32}(0)
    check(ccall((:ccal_detect_corners, LIB), Int32,
                (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Int64, Int64, Int32, Int32, Ptr{Float64}, Ref{Int32}, Ptr{Float64}),
                c.h, gray, rows, cols, 1, rows, n1, n2, corners, found, C_NULL))
    found[] == 0 && return missing
    return reshape([(corners[1, i], corners[2, i]) for i in 1:n1*n2], n1, n2)
end

# the images of one calibration (one size) in one call: three in flight on the device (include/ccal.h ccal_detect_corners_many)
function detect_corners_many(c::Context, grays::Vector{Matrix{UInt8}}, n_corners)
    isempty(grays) && return []
    # the C side reads rows x cols bytes from every pointer: images of another size go one by one
    all(size(g) == size(grays[1]) for g in grays) || return [detect_corners(c, g, n_corners) for g in grays]
    rows, cols = size(grays[1]); n1, n2 = n_corners
    outs = [zeros(Float64, 2, n1 * n2) for _ in grays]; found = zeros(Int32, length(grays))
    GC.@preserve grays outs begin
        ip = Ptr{Cvoid}[pointer(g) for g in grays]; op = Ptr{Cvoid}[pointer(o) for o in outs]
        check(ccall((:ccal_detect_corners_many, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32, Int32, Int32, Int64, Int64, Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Int32}),
                    c.h, ip, length(grays), rows, cols, 1, rows, n1, n2, op, found))
    end
    return [f == 0 ? missing : reshape([(o[1, i], o[2, i]) for i in 1:n1*n2], n1, n2) for (o, f) in zip(outs, found)]
end

# ---- many-image maps: image n through pose n, one call (include/ccal.h ccal_project_many / ccal_unproject_many) ----
function project_many(c::Context, xyz::Array{Float64}, f, cc, k, rvecs::Matrix{Float64}, tvecs::Matrix{Float64}, checker_size)
    N = size(rvecs, 2); per_image = ndims(xyz) == 3; M = size(xyz, 2); out = zeros(Float64, 2, M, N)
    check(ccall((:ccal_project_many, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                c.h, xyz, M, N, per_image ? 1 : 0, Float64[f...], Float64[cc...], Float64(k), rvecs, tvecs, Float64(checker_size), out))
    return out
end

function unproject_many(c::Context, rowcol::Array{Float64,3}, f, cc, k, rvecs::Matrix{Float64}, tvecs::Matrix{Float64}, checker_size)
    M, N = size(rowcol, 2), size(rowcol, 3); out = zeros(Float64, 3, M, N)
    check(ccall((:ccal_unproject_many, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                c.h, rowcol, M, N, Float64[f...], Float64[cc...], Float64(k), rvecs, tvecs, Float64(checker_size), out))
    return out
end

# ---- the reference's public surface on top of the calls above (src/CameraCalibrations.jl:18 exports
#      fit, Calibration, RowCol, XYZ, rectification) ------------------------------------------------------
const RowCol = SVector{2, <: Real}      # src/meta.jl:9
const XYZ = SVector{3, <: Real}         # src/meta.jl:15

"""
    Calibration
Mirror of the reference struct (src/meta.jl:17-25) holding numbers instead of composed closures: the maps are
evaluated by the device library.  `c(xyz, i)`, `c(rc, i)`, `c(x)` and `rectification(c, i)` behave as in the
reference (src/meta.jl:82-103); a `Matrix` argument ([3,M] / [2,M]) maps M points in one call.
"""
struct Calibration
    f::SVector{2, Float64}               # intrinsic.linear diagonal (frow, fcol)
    c::SVector{2, Float64}               # intrinsic.translation (crow, ccol)
    rvecs::Matrix{Float64}               # [3,N] RotationVec per image
    tvecs::Matrix{Float64}               # [3,N]
    checker_size::Float64                # scale = LinearMap(I / checker_size), src/buildcalibrations.jl:4
    k::Float64
    files::Vector{String}
    ctx::Context
end

(c::Calibration)(xyz::XYZ, i::Int) = SVector{2}(project(c.ctx, reshape(Float64[xyz...], 3, 1), c.f, c.c, c.k, c.rvecs[:, i], c.tvecs[:, i], c.checker_size))   # src/meta.jl:88
(c::Calibration)(rc::RowCol, i::Int) = SVector{3}(unproject(c.ctx, reshape(Float64[rc...], 2, 1), c.f, c.c, c.k, c.rvecs[:, i], c.tvecs[:, i], c.checker_size))  # src/meta.jl:82
(c::Calibration)(pts::Matrix{Float64}, i::Int) = size(pts, 1) == 3 ? project(c.ctx, pts, c.f, c.c, c.k, c.rvecs[:, i], c.tvecs[:, i], c.checker_size) :
                                                                    unproject(c.ctx, pts, c.f, c.c, c.k, c.rvecs[:, i], c.tvecs[:, i], c.checker_size)
# a vector of points maps in ONE device call (a loop over points like src/buildcalibrations.jl:53-58 should be written
# `c(pts, i)` or `c.(Ref(pts), i)`: a point-wise call costs a launch and a completion wait, ~20 us)
(c::Calibration)(pts::AbstractVector{<:XYZ}, i::Int) = [SVector{2}(col) for col in eachcol(c(Float64[p[d] for d in 1:3, p in pts], i))]
(c::Calibration)(pts::AbstractVector{<:RowCol}, i::Int) = [SVector{3}(col) for col in eachcol(c(Float64[p[d] for d in 1:2, p in pts], i))]
# all images at once, image n through pose n: [3,M,N] or one [3,M] template -> [2,M,N]; [2,M,N] -> [3,M,N]
map_all(c::Calibration, x::Array{Float64}) = size(x, 1) == 3 ? project_many(c.ctx, x, c.f, c.c, c.k, c.rvecs, c.tvecs, c.checker_size) :
                                                               unproject_many(c.ctx, x, c.f, c.c, c.k, c.rvecs, c.tvecs, c.checker_size)

function extrinsic_index(c::Calibration)                                    # src/meta.jl:90-93
    i = findfirst(contains(r"extrinsic"), c.files)
    i === nothing && throw(ArgumentError("no file name matches r\"extrinsic\""))   # (the reference fails on its `nothing` index)
    return i
end
(c::Calibration)(x) = c(x, extrinsic_index(c))

rectification(c::Calibration, i) = rc -> c(rc, i)[SVector(1, 2)]            # src/meta.jl:99-103
rectification(c::Calibration) = rectification(c, extrinsic_index(c))

"""
    fit_corners(imgpointss, n_corners, checker_size, sz; files, aspect, with_distortion, inverse_samples, device)
`fit` from already-detected corners (detection stays with OpenCV.jl or `detect_corners`): everything after
`_detect_corners` in src/detect_fit.jl:63-72 and src/buildcalibrations.jl:12-26.  Returns `(c, ϵ)`.
"""
function fit_corners(imgpointss, n_corners, checker_size, sz; files = ["$i.png" for i in 1:length(imgpointss)], aspect = 1,
                     with_distortion = true, inverse_samples = 100, device = 0, ctx = Context(device))
    @assert !isempty(imgpointss) "No checkers were detected in any of the images, perhaps try a different `n_corners`."   # src/detect_fit.jl:66
    n1, n2 = n_corners
    objpoints = [SVector{3, Float64}(i - 1, j - 1, 0) for i in 1:n1, j in 1:n2, _ in 1:1]          # src/detect_fit.jl:69
    m = fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect; device)
    N, P = length(imgpointss), n1 * n2
    c = Calibration(SVector(m.frow, m.fcol), SVector(m.crow, m.ccol), reduce(hcat, m.Rs), reduce(hcat, m.ts), Float64(checker_size),
                    m.k, collect(String, files), ctx)
    obj = Float64[(p - 1) % n1 * (d == 1) + (p - 1) ÷ n1 * (d == 2) for d in 1:3, p in 1:P] .* checker_size   # src/buildcalibrations.jl:15
    img = Array{Float64}(undef, 2, P, N)
    for (i, mm) in enumerate(imgpointss), (p, rc) in enumerate(vec(mm))
        img[1, p, i] = rc[1]; img[2, p, i] = rc[2]
    end
    inv_pts = rand(2, inverse_samples, N) .* (Float64[sz...] .- 1) .+ 1                                   # src/buildcalibrations.jl:54
    ϵ = calculate_errors(c.ctx, n_corners, obj, img, c.f, c.c, c.k, c.rvecs, c.tvecs, checker_size, inv_pts)
    return c, ϵ
end

"""
    fit(files, n_corners, checker_size; aspect, with_distortion, inverse_samples, load_gray)
src/buildcalibrations.jl:12-26 with corner detection on the device.  `load_gray(file) -> Matrix{UInt8}` loads an image (the
reference's `fit` has no such keyword: it reads the files itself through FileIO / OpenCV.jl, which this binding does not
depend on); images without a full board are skipped like the reference's `missing`, src/detect_fit.jl:65.  One device
context serves the detection, the fit's error metrics and the returned `Calibration`.
"""
function fit(files, n_corners, checker_size; aspect = 1, with_distortion = true, inverse_samples = 100, device = 0, load_gray)
    ctx = Context(device)
    found = Tuple{String, Any}[]
    sz = nothing
    for file in unique(files)
        gray = load_gray(file)
        sz === nothing && (sz = size(gray))                         # size(FileIO.load(first(files))), src/detect_fit.jl:64
        corners = detect_corners(ctx, gray, n_corners)
        ismissing(corners) || push!(found, (file, [SVector(rc...) for rc in corners]))
    end
    return fit_corners(last.(found), n_corners, checker_size, sz; files = first.(found), aspect, with_distortion, inverse_samples, device, ctx)
end

end # module
