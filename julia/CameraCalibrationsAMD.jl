# CameraCalibrationsAMD.jl -- thin `ccall` binding of libccal_hip.so (include/ccal.h).
#
# UNTESTED IN THIS REPOSITORY: the build image has no Julia.  The Python mirror
# (cameracalibrations_amd/solver.py, tested on MI355X) makes exactly the same calls with the
# same array layouts; keep the two in step.
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

end # module
