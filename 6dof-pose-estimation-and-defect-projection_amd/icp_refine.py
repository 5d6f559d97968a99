"""Host mirror of the reference's ICP refinement functions (src/pose_estimation.py), same
names, positional signatures, return values and input mutations, with every
registration_icp call running on the GPU through libpedp_hip.so.

    refine_registration          pose_estimation.py:505-522
    improve_result               :547-622   (randomised restarts, global numpy RNG)
    predict_z_axis_adjustment    :624-683   (adaptive camera-z line search)
    refine_pose_with_icp         :749-822   (orchestrator called by run.py:99)
    transform_object             :406-409
    preprocess_target            :141-183   (random subsample to max_pcd; keeps normals)
    preprocess_source            :186-268   (voxel grid, table plane, DBSCAN, outlier filter and
                                             normals on the GPU through pedp_hip.cloud_ops)
    execute_global_registration  :467-503   (FPFH + RANSAC global registration on the GPU)
    run_icp                      :524-545
    determine_pose               :686-747   (icp=False: the FoundationPose chain; icp=True: run_icp)

Units are millimetres, transformations are 4x4 float64; `*.transformation` of a result maps
scene -> model, exactly like the reference's RegistrationResult.
"""
import copy
import logging

import numpy as np

from . import registration as reg
from .geometry import (KDTreeSearchParamHybrid, PointCloud, RegistrationResult, TriangleMesh, as_holder, clone, normals_of,
                       points_of, rigid)


def _copy_parameters(tree):
    """copy.deepcopy of the parameter tree (dicts, lists, numbers, strings) without deepcopy's bookkeeping: the reference
    copies its parameters once per call (pose_estimation.py:561, :766) and the generic copy was 0.2 ms of every frame."""
    if isinstance(tree, dict):
        return {k: _copy_parameters(v) for k, v in tree.items()}
    if isinstance(tree, list):
        return [_copy_parameters(v) for v in tree]
    if isinstance(tree, (int, float, str, bool, type(None))):
        return tree
    return copy.deepcopy(tree)


def transform_object(pcd, transformation):
    """Deep copy moved by `transformation` (pose_estimation.py:406-409).  Our own holders are built from the moved arrays
    directly -- copy-then-transform writes the points twice -- with the same arithmetic as their transform(); a point
    cloud forms them when it is first read."""
    if type(pcd) is PointCloud:
        return PointCloud.moved_copy(pcd, transformation)   # formed when first read (run.py:99 never reads it)
    if type(pcd) is TriangleMesh:
        T = np.asarray(transformation, dtype=np.float64)
        moved = TriangleMesh(rigid(pcd.vertices, T), np.array(pcd.triangles))
        moved.vertex_normals = rigid(pcd.vertex_normals, T, rotate_only=True) if len(pcd.vertex_normals) else np.array(pcd.vertex_normals)
        moved.triangle_normals = rigid(pcd.triangle_normals, T, rotate_only=True) if len(pcd.triangle_normals) else np.array(pcd.triangle_normals)
        return moved
    moved = clone(pcd)
    moved.transform(transformation)
    return moved


def refine_registration(source, target, transformation, param):
    """One full point-to-plane ICP with Open3D's default criteria (pose_estimation.py:505-522)."""
    return reg.registration_icp(source, target, param["refine_registration"]["distance_threshold"],
                                transformation, reg.TransformationEstimationPointToPlane())


def estimate_normals(pcd, params):
    """pose_estimation.py:301-306: hybrid search radius 2, at most 5 neighbours (params unused there too)."""
    pcd.estimate_normals(search_param=KDTreeSearchParamHybrid(radius=2, max_nn=5))
    return pcd


def compute_average_normal(pcd):
    """pose_estimation.py:314-321: mean normal of the cloud on a 10-unit voxel grid, normalised."""
    normals = np.asarray(pcd.voxel_down_sample(voxel_size=10).normals)     # (voxel_down_sample leaves its input alone)
    average_normal = np.mean(normals, axis=0)
    return average_normal / np.linalg.norm(average_normal)


def preprocess_target(pcd, param):
    """Subsample the model cloud to `max_pcd` points with the global numpy RNG
    (np.random.choice(..., replace=False), pose_estimation.py:159-169), then re-estimate its normals
    like the reference (:174; radius 2, max_nn 5 -- a model sampled more coarsely than ~1 unit gets
    (0, 0, 1) everywhere, in Open3D too).  `"keep_normals": True` in the preprocess_target section
    (not a reference key) keeps the model's own normals instead.  FPFH features (:175-179) are computed
    when the section names `fpfh_radius` / `fpfh_max_nn` (the reference's configuration always does; only
    the global-registration path reads them), else the slot is None."""
    section = param["preprocess_target"]
    cap = section["max_pcd"]
    pts, nrm = points_of(pcd), normals_of(pcd)
    if len(pts) > cap:
        keep = np.random.choice(len(pts), cap, replace=False)
        out = PointCloud.adopt(pts[keep], None if nrm is None else nrm[keep])
        if hasattr(pcd, "has_colors") and pcd.has_colors():
            out.colors = np.asarray(pcd.colors)[keep]
    else:
        logging.info(f":: Point cloud already has less than or exactly {cap} points.")
        out = pcd
    if section.get("keep_normals"):
        if normals_of(out) is None:
            raise RuntimeError("preprocess_target: keep_normals is set but the model cloud carries no normals")
    else:
        if not isinstance(out, PointCloud):  # an Open3D cloud: estimate on the GPU and write the normals back
            holder = as_holder(out)
            estimate_normals(holder, section)
            try:
                import open3d as o3d  # only reachable when the caller works with Open3D objects

                out.normals = o3d.utility.Vector3dVector(holder.normals)
            except ImportError:
                out = holder
        else:
            estimate_normals(out, section)
    return out, fpfh_of(out, section)


def fpfh_of(pcd, section):
    """compute_fpfh_feature with the section's hybrid search (pose_estimation.py:175-180, :255-260); None when the
    section does not ask for features."""
    if pcd is None or "fpfh_radius" not in section:
        return None
    return reg.compute_fpfh_feature(pcd, KDTreeSearchParamHybrid(radius=section["fpfh_radius"], max_nn=section["fpfh_max_nn"]))


def perform_plane_segmentation(pcd, param):
    """pose_estimation.py:323-329."""
    return pcd.segment_plane(distance_threshold=param["distance_threshold"], ransac_n=3,
                             num_iterations=param["num_iterations"])


def flip_plane_normal_if_needed(plane_model, average_normal):
    """pose_estimation.py:342-359: orient the plane like the cloud's average normal."""
    a, b, c, d = plane_model
    plane_normal = np.array([a, b, c], dtype=np.float64)
    plane_normal /= np.linalg.norm(plane_normal)
    if np.dot(plane_normal, average_normal) < 0:
        plane_normal = -plane_normal
        plane_model = [-a, -b, -c, -d]
        logging.info(":: Plane normal was flipped to match the majority of normals.")
    return plane_model, plane_normal


def remove_plane(pcd, inliers):
    return pcd.select_by_index(inliers, invert=True)


def remove_points_below_plane(pcd, plane_model):
    """Keep the points with signed distance <= 0 (pose_estimation.py:366-378); like the reference the
    result carries points only."""
    a, b, c, d = plane_model
    points = points_of(pcd)
    distances = (a * points[:, 0] + b * points[:, 1] + c * points[:, 2] + d) / np.sqrt(a ** 2 + b ** 2 + c ** 2)
    return PointCloud.adopt(points[distances <= 0])


def background_removal(pcd, background_pcd, threshold=10):
    """pose_estimation.py:380-392 always returns its input: the freshly made result cloud is tested
    for emptiness before it is filled (:386-388).  Kept, because it decides which points reach ICP."""
    return pcd


def filter_largest_cluster(pcd, eps=10, min_points=10):
    """DBSCAN, keep the largest cluster (pose_estimation.py:270-299); None when everything is noise."""
    labels = np.array(pcd.cluster_dbscan(eps=eps, min_points=min_points, print_progress=True))
    unique_labels, counts = np.unique(labels, return_counts=True)
    valid_labels = unique_labels[unique_labels != -1]
    if len(valid_labels) == 0:
        print("No valid clusters found.")
        return None
    largest = valid_labels[np.argmax(counts[unique_labels != -1])]
    return pcd.select_by_index(np.where(labels == largest)[0])


def remove_statistical_outliers(pcd, nb_neighbors=20, std_ratio=1.0):
    clean, _ = pcd.remove_statistical_outlier(nb_neighbors=nb_neighbors, std_ratio=std_ratio)
    return clean


def preprocess_source(pcd, background, param, i=0):
    """Scene preprocessing (pose_estimation.py:186-268): voxel grid, RANSAC table plane, half-space
    cut, (background removal | plane removal), largest DBSCAN cluster, statistical outlier filter.
    The point-cloud operations run on the GPU (pedp_hip.cloud_ops).  Returns (processed, filtered,
    fpfh) with filtered is processed, as in the reference.

    FPFH features (:254-260) are computed when the section names fpfh_radius / fpfh_max_nn (only the
    global-registration path reads them), else the slot is None (i == 0) or 0 (i > 0).  Not built:
    param['mesh'] (Poisson re-meshing, :240-245) raises NotImplementedError.
    A param dict without a 'preprocess_source' section means "already preprocessed": the cloud is
    passed through (the tracking-frame mutation down_sample = 5 is still applied)."""
    if "preprocess_source" not in param:
        if i > 0:
            param.setdefault("preprocess_source", {})["down_sample"] = 5
        return pcd, pcd, None
    params = param["preprocess_source"]
    if i > 0:
        params["down_sample"] = 5
    pcd, background = as_holder(pcd), as_holder(background)  # Open3D inputs: the GPU methods, not Open3D's
    # Every branch run.py can take except param['mesh'] runs as ONE library call with the scene on the device between
    # the stages (pedp_preprocess_source_ex): the same kernels and rules as the calls below, bit for bit the same cloud
    # (tests/test_cloudops_gpu.py).  run.py always passes reader.background (run.py:99-101, :154-156) and logs at INFO
    # (run.py:252, :260): the background cloud changes nothing a caller can see -- its down-sampled copy and normals
    # (:204, :252) are locals of the reference's function, without param['box'] the cut it could take part in is
    # overwritten (:232-236) and with it background_removal returns its input (:386-388) -- so its work is skipped; the
    # two INFO lines (:220, :356) get the average normal and the refit plane back from the call.  A frame that leaves no
    # cluster, an undecidable plane flip and param['debug_vis'] (the reference's screenshots, :206-268) go through the
    # steps, whose behaviour is then the reference's to the letter.
    box = bool(param.get("box"))
    info = logging.getLogger().isEnabledFor(logging.INFO)
    if not param.get("mesh") and not param.get("debug_vis") and not _FORCE_STEPS:
        from . import cloud_ops

        src_pts = pcd._dev_points if getattr(pcd, "_dev_points", None) is not None else points_of(pcd)   # (a holder's device twin, if it has one)
        plane = params["plane_removal"]
        res = cloud_ops.preprocess_source_fused(src_pts, params["down_sample"], plane["distance_threshold"], plane["num_iterations"],
                                                first_frame=(i == 0), box=box, report=(info or box))
        if res[3] == 0:
            flipped, average_normal = None, np.array([1, 1, 1], dtype=float)
            if info or box:           # the reference's own numpy expressions (:217-222, :342-359) on the call's report
                rep = res[4]
                if i == 0:
                    average_normal = rep["mean_normal"] / np.linalg.norm(rep["mean_normal"])
                plane_normal = np.array(rep["plane_model"][:3], dtype=np.float64)
                plane_normal /= np.linalg.norm(plane_normal)
                flipped = bool(np.dot(plane_normal, average_normal) < 0)
            if not box or flipped == rep["flipped"]:
                if i == 0 and info:
                    logging.info(f":: Average Normal for Source = {average_normal}")
                if flipped:
                    logging.info(":: Plane normal was flipped to match the majority of normals.")
                source_processed = PointCloud.adopt(res[0], res[1])
                return source_processed, source_processed, (fpfh_of(source_processed, params) if i == 0 else 0)
    if background is not None:
        background = background.voxel_down_sample(voxel_size=params["down_sample"] * 2)
    pcd_down = pcd.voxel_down_sample(voxel_size=params["down_sample"])
    plane_model, inliers = perform_plane_segmentation(pcd_down, params["plane_removal"])
    # Without param['box'] the half-space cut below is thrown away (the reference overwrites it, :232-236), and
    # with it the only reader of the flipped plane and of the average normal -- apart from two log lines.
    # They are worked out when something can see them: the box branch, or INFO logging.
    seen = box or info
    average_normal = np.array([1, 1, 1], dtype=float)
    if i == 0:
        estimate_normals(pcd_down, params)      # (kept either way: these normals orient the final ones)
        if seen:
            average_normal = compute_average_normal(pcd_down)
            logging.info(f":: Average Normal for Source = {average_normal}")
    if seen:
        plane_model, _ = flip_plane_normal_if_needed(plane_model, average_normal)
    if box:
        source_processed = background_removal(remove_points_below_plane(pcd_down, plane_model), background)
    else:
        source_processed = remove_plane(pcd_down, inliers)
    if param.get("mesh"):
        raise NotImplementedError("preprocess_source: param['mesh'] (Poisson re-meshing) is not part of this build")
    source_processed = filter_largest_cluster(source_processed)
    source_processed = remove_statistical_outliers(source_processed, nb_neighbors=75, std_ratio=0.01)
    source_fpfh = 0
    if i == 0:
        if background is not None:
            estimate_normals(background, params)
        estimate_normals(source_processed, params)
        source_fpfh = fpfh_of(source_processed, params)
    return source_processed, source_processed, source_fpfh


_FORCE_STEPS = False   # tests: preprocess_source through the single operations

Z_LOOKAHEAD = 3   # probes tried ahead per batch: 2^3 - 1 = 7 start poses share the launches of one (4: 15 poses per batch measured slower)


def _z_move(offset, step, heading, better, max_adjustment):
    """Where the line search goes after a probe that was / was not an improvement (pose_estimation.py:661-675):
    keep walking, or turn round and halve; clamp to +-max_adjustment (step / 1.25, turn)."""
    if not better:
        heading = -heading
        step = step / 2
    offset += step * heading
    if abs(offset) > max_adjustment:
        offset = max_adjustment * np.sign(offset)
        step = step / 1.25
        heading = -heading
    return offset, step, heading


def predict_z_axis_adjustment(source, target, initial_fp_transformation, param, max_adjustment=50,
                              initial_step=10):
    """Adaptive 1-D search for the camera-z offset that maximises ICP fitness
    (pose_estimation.py:624-683).  Each probe is a single-iteration point-to-plane ICP started
    from inv(T) with T[2,3] lowered by the probe offset.  Walk in the current direction while
    the probe improves (fitness, then lower rmse); otherwise turn round and halve the step;
    clamp to +-max_adjustment (step / 1.25, turn); stop below 0.1 mm or above 0.95 fitness.

    Where the search goes next depends only on WHETHER a probe improved, so the offsets of the next
    Z_LOOKAHEAD probes are known for both answers: the 2^k - 1 candidate poses run as one batch and
    the results are then taken along the path the real answers pick -- the same probes in the same
    order with the same numbers as one by one (the rest are discarded)."""
    radius = param["refine_registration"]["distance_threshold"]
    d_src, d_tgt = reg.upload(source), reg.upload(target)  # one upload for all probes
    one_iteration = reg.ICPConvergenceCriteria(max_iteration=1)
    plane = reg.TransformationEstimationPointToPlane()
    best_adjustment, best_fitness, best_rmse = 0, 0, float("inf")
    offset, step, heading = 0, initial_step, 1
    done = False
    while abs(step) >= 0.1 and not done:
        # the tree of probes the next Z_LOOKAHEAD answers can lead to; node = (offset, step, heading)
        nodes = {(): (offset, step, heading)}
        frontier = [()]
        for _ in range(Z_LOOKAHEAD - 1):
            grown = []
            for path in frontier:
                for better in (True, False):
                    nxt = _z_move(*nodes[path], better, max_adjustment)
                    if abs(nxt[1]) >= 0.1:               # a path that has run out of step probes nothing more
                        nodes[path + (better,)] = nxt
                        grown.append(path + (better,))
            frontier = grown
        paths = list(nodes)
        probes = np.repeat(np.asarray(initial_fp_transformation, dtype=np.float64)[None], len(paths), axis=0)
        probes[:, 2, 3] -= [nodes[path][0] for path in paths]
        starts = np.linalg.inv(probes)                       # one call; the same LAPACK solve per matrix as one by one
        results = dict(zip(paths, reg.registration_icp_batch(d_src, d_tgt, [radius] * len(paths), starts, plane, one_iteration)))
        path = ()
        while path in results:
            res = results[path]
            offset, step, heading = nodes[path]
            better = res.fitness > best_fitness or (res.fitness == best_fitness and res.inlier_rmse < best_rmse)
            if better:
                best_adjustment, best_fitness, best_rmse = offset, res.fitness, res.inlier_rmse
            offset, step, heading = _z_move(offset, step, heading, better, max_adjustment)
            if best_fitness > 0.95:
                done = True
                break
            if abs(step) < 0.1:
                break
            path = path + (better,)
    logging.info(f":: Best z-axis adjustment: {best_adjustment:.2f}mm, Fitness: {best_fitness:.4f}, "
                 f"RMSE: {best_rmse:.4f}")
    return best_adjustment, best_fitness, best_rmse


SPECULATION = 8   # restarts tried at once (the library keeps up to eight registrations in flight)


def improve_result(source_processed, original_target_processed, current_result, parameter, trace=None):
    """Up to 50 randomised ICP restarts around the best scene->model transformation so far
    (pose_estimation.py:547-622).

    Reference behaviours kept because they change the returned numbers:
      * `current_result` may be a bare 4x4 (tracking frames, run.py:168): treated as
        fitness 0.8 / rmse 3.0 (:564-569);
      * the search starts from inv(current_result.transformation) (:572);
      * RNG order per restart: 1 x uniform(0.8, 1.2), 3 x uniform(-0.01, 0.01), 1 x
        uniform(-x, x, 3) from the global numpy RNG;
      * the distance threshold is scaled IN PLACE on a shallow copy, so it random-walks
        across restarts (:580-582);
      * a result with zero fitness or rmse widens the translation noise by 0.25 (:609);
      * exceptions in a restart are logged and skipped (:610-611).

    Restarts are tried several at a time.  A restart's random draws depend on nothing but the RNG
    state and the noise width, and its start pose on nothing but the best pose so far; so the next
    n restarts are drawn ahead under the assumption that none of them changes the best pose or the
    noise width, and they run as ONE batch on the GPU.  The results are then taken in order: up to
    and including the first restart that does change something (an improvement, or an invalid
    result), the sequence is exactly the reference's; the restarts behind it are discarded and the
    RNG is put back to the state it had after that restart.  Returned numbers, the threshold walk
    and the RNG state on return are those of the one-by-one loop.  `trace` (a list, optional)
    receives (threshold, fitness, rmse) of every restart that counted."""
    settings = _copy_parameters(parameter)
    if not hasattr(current_result, "fitness") or current_result.fitness is None:
        seed = RegistrationResult(current_result)
        seed.fitness, seed.inlier_rmse = 0.8, 3.0
        current_result = seed
    best_fitness, best_rmse = current_result.fitness, current_result.inlier_rmse
    best_T = np.linalg.inv(current_result.transformation)
    want_fitness = settings["run_icp"]["fitness_threshold"]
    want_rmse = settings["run_icp"]["rmse_threshold"]
    d_src, d_tgt = reg.upload(source_processed), reg.upload(original_target_processed)
    plane = reg.TransformationEstimationPointToPlane()
    spread, rounds, ahead = 0.1, 0, 1
    logging.info(":: Additional refinements")

    def unfinished():
        return rounds < 50 and (best_fitness < want_fitness or best_rmse > want_rmse)

    while unfinished():
        # ---- draw the next `n` restarts as the one-by-one loop would if nothing changed in between
        n = min(ahead, 50 - rounds)
        rng_before = np.random.get_state() if n > 1 else None          # (a copy of the 624-word state: only where it can be needed)
        radius = settings["refine_registration"]["distance_threshold"]
        radii, starts, rng_after = [], [], []
        for k_ahead in range(n):
            radius = radius * np.random.uniform(0.8, 1.2)
            wobble = np.eye(4)
            wobble[:3, :3] = reg.get_rotation_matrix_from_xyz([np.random.uniform(-0.01, 0.01) for _ in range(3)])
            wobble[:3, 3] = np.random.uniform(-spread, spread, 3)
            radii.append(radius)
            starts.append(wobble @ best_T)
            rng_after.append(np.random.get_state() if k_ahead + 1 < n else None)   # (the state behind the last draw is the live one)
        try:
            results = reg.registration_icp_batch(d_src, d_tgt, radii, starts, plane)
        except Exception as exc:
            if n > 1:  # a batch cannot say which restart failed: take this stretch one by one
                np.random.set_state(rng_before)
                ahead = 1
                continue
            results = [exc]
        # ---- take the results in order, up to the first one that changes the search state
        taken, changed = 0, False
        for k, res in enumerate(results):
            taken, rounds = k + 1, rounds + 1
            settings["refine_registration"]["distance_threshold"] = radii[k]   # the shallow-copy walk of the reference
            if isinstance(res, Exception):  # same contract as the reference: a failed restart is skipped
                logging.info(f":: Error in refinement iteration {rounds}: {res}. Skipping this iteration.")
            elif res.fitness > 0 and res.inlier_rmse > 0:
                if trace is not None:
                    trace.append((radii[k], res.fitness, res.inlier_rmse))
                if res.fitness > best_fitness or (res.fitness == best_fitness and res.inlier_rmse < best_rmse):
                    best_fitness, best_rmse, best_T = res.fitness, res.inlier_rmse, res.transformation
                    logging.info(f":: Improved result: Fitness = {best_fitness:.4f}, RMSE = {best_rmse:.4f}")
                    changed = True
            else:
                if trace is not None:
                    trace.append((radii[k], res.fitness, res.inlier_rmse))
                logging.info(f":: Iteration {rounds} produced an invalid result. Skipping.")
                spread += 0.25
                changed = True
            if changed:
                break
        if taken < n:
            np.random.set_state(rng_after[taken - 1])
        # look further ahead while nothing happens, start over after a change
        ahead = 1 if changed else min(SPECULATION, 2 * ahead)
    logging.info(f":: Total iterations: {rounds}")
    out = RegistrationResult(best_T)
    out.fitness, out.inlier_rmse = best_fitness, best_rmse
    return out


def refine_pose_with_icp(source, target, background, initial_fp_transformation, parameters):
    """FoundationPose estimate -> z search -> randomised ICP refinement (pose_estimation.py:
    749-822; caller run.py:99).  Mutations kept: both clouds are painted (:769-770) and the
    caller's `initial_fp_transformation[2, 3]` receives the z adjustment in place (:789),
    which run.py:104 relies on.  Returns (model moved into the scene, result, z adjustment,
    preprocessed target)."""
    param = _copy_parameters(parameters)
    if hasattr(source, "paint_uniform_color"):
        source.paint_uniform_color([1, 0, 0])
    if hasattr(target, "paint_uniform_color"):
        target.paint_uniform_color([0, 0, 1])
    target_processed, _ = preprocess_target(target, param)
    source_processed, _, _ = preprocess_source(source, background, param)

    z_adjustment, fitness, rmse = predict_z_axis_adjustment(source_processed, target_processed,
                                                            initial_fp_transformation, param)
    initial_fp_transformation[2, 3] += z_adjustment
    logging.info(f":: Predicted Z-axis adjustment: {z_adjustment:.2f}mm")

    start = RegistrationResult(initial_fp_transformation)
    start.fitness, start.inlier_rmse = fitness, rmse
    best = improve_result(source_processed, target_processed, start, param)
    model_in_scene = np.linalg.inv(best.transformation)
    if logging.getLogger().isEnabledFor(logging.INFO):  # printing the matrix is 0.2 ms whether or not anyone reads it
        logging.info(f"-- Final Results\n:: Refine registration results: Inlier_rmse: {best.inlier_rmse:.4f}, "
                     f"Fitness: {best.fitness:.4f}\n:: Final Transformation Matrix:\n{model_in_scene}")
    target_transformed = transform_object(target, model_in_scene)
    return target_transformed, best, z_adjustment, target_processed


def execute_global_registration(source_processed, target_processed, source_fpfh, target_fpfh, param):
    """RANSAC on feature matches with the reference's three checkers (pose_estimation.py:467-503)."""
    params = param["execute_global_registration"]
    return reg.registration_ransac_based_on_feature_matching(
        source_processed, target_processed, source_fpfh, target_fpfh, False, params["distance_threshold"],
        reg.TransformationEstimationPointToPoint(False), 3,
        [reg.CorrespondenceCheckerBasedOnEdgeLength(params["correspondence_checkers"][0]["value"]),
         reg.CorrespondenceCheckerBasedOnDistance(params["distance_threshold"]),
         reg.CorrespondenceCheckerBasedOnNormal(params["angle_threshold"])],
        reg.RANSACConvergenceCriteria(params["ransac_criteria"]["iterations"], params["ransac_criteria"]["confidence"]))


def run_icp(source_processed, target_processed, source_fpfh, target_fpfh, param):
    """Global registration, then one point-to-plane refinement from it (pose_estimation.py:524-545).
    Returns (result_icp, result_ransac)."""
    result_ransac = execute_global_registration(source_processed, target_processed, source_fpfh, target_fpfh, param)
    result_icp = refine_registration(source_processed, target_processed, result_ransac.transformation, param)
    return result_icp, result_ransac


MAX_GLOBAL_ATTEMPTS = 1000   # the reference loops without a bound (pose_estimation.py:702-710)


def determine_pose(source, target, background, initial_fp_transformation, parameters, icp=False):
    """pose_estimation.py:686-747.  icp=False (what run.py's callers use) is the chain of
    refine_pose_with_icp: preprocess, z search, in-place z adjustment, improve_result.  icp=True starts
    from run_icp -- FPFH features, RANSAC global registration, one refinement -- repeated until the
    refined result meets run_icp's fitness / rmse thresholds (:698-710; the reference repeats without a
    bound, here MAX_GLOBAL_ATTEMPTS attempts raise), then inverts the transformation (:711) and hands it
    to improve_result like the other branch.  Returns (model moved into the scene, result, z adjustment,
    preprocessed target)."""
    if not icp:
        return refine_pose_with_icp(source, target, background, initial_fp_transformation, parameters)
    param = _copy_parameters(parameters)
    if hasattr(source, "paint_uniform_color"):
        source.paint_uniform_color([1, 0, 0])
    if hasattr(target, "paint_uniform_color"):
        target.paint_uniform_color([0, 0, 1])
    target_processed, target_fpfh = preprocess_target(target, param)
    source_processed, _, source_fpfh = preprocess_source(source, background, param)
    if target_fpfh is None or source_fpfh is None:
        raise KeyError("determine_pose(icp=True): the preprocess_target / preprocess_source sections need fpfh_radius and fpfh_max_nn")
    result_icp, result_ransac = run_icp(source_processed, target_processed, source_fpfh, target_fpfh, param)
    logging.info(f"-- Initial Attempt\n:: Global registeration results: Inlier_rmse: {result_ransac.inlier_rmse:.4f}, "
                 f"Fitness: {result_ransac.fitness:.4f}\n:: Refine registeration results: Inlier_rmse: "
                 f"{result_icp.inlier_rmse:.4f}, Fitness: {result_icp.fitness:.4f}")
    attempts = 1
    while result_icp.fitness < param["run_icp"]["fitness_threshold"] or result_icp.inlier_rmse > param["run_icp"]["rmse_threshold"]:
        if attempts >= MAX_GLOBAL_ATTEMPTS:
            raise RuntimeError(f"determine_pose(icp=True): {attempts} global registrations did not reach the run_icp thresholds")
        result_icp, result_ransac = run_icp(source_processed, target_processed, source_fpfh, target_fpfh, param)
        logging.info(f"-- Attempt {attempts}\n:: Global registeration results: Inlier_rmse: {result_ransac.inlier_rmse:.4f}, "
                     f"Fitness: {result_ransac.fitness:.4f}\n:: Refine registeration results: Inlier_rmse: "
                     f"{result_icp.inlier_rmse:.4f}, Fitness: {result_icp.fitness:.4f}")
        attempts += 1
    result_icp.transformation = np.linalg.inv(result_icp.transformation)
    best = improve_result(source_processed, target_processed, result_icp, param)
    model_in_scene = np.linalg.inv(best.transformation)
    logging.info(f"-- Final Results\n:: Refine registration results: Inlier_rmse: {best.inlier_rmse:.4f}, "
                 f"Fitness: {best.fitness:.4f}\n:: Final Transformation Matrix:\n{model_in_scene}")
    return transform_object(target, model_in_scene), best, 0, target_processed
