// integration_snippet.cpp — the edit INTEGRATION.md §A describes, as one translation unit: what a maintainer of the
// reference adds to pcl_preprocessing/src/pc_preprocessing_main.cpp to put the MI355X path behind its node. Not
// built into anything here (no ROS in this image); tests/test_integration_snippet.py checks that it still compiles
// against include/cloudmerge.h (-fsyntax-only, tests/ros_stub/ standing in for the ROS headers) and that the block
// between the two markers is, character for character, the code block INTEGRATION.md prints.
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_listener.h>

// the reference's own globals these lines use (Parameter.h:27-35, pc_preprocessing_main.h:58-63, :518)
extern float voxel_size, roi_mid, roi_length, roi_width, roi_z_min, roi_z_max;
extern int points_per_voxel;
extern tf::StampedTransform front_right_stf, front_left_stf, rear_right_stf, rear_left_stf, top_middle_stf, front_middle_stf;
extern ros::Publisher voxelpub;

// [snippet-begin]
#include <cloudmerge.h>
static cm_ctx* g_cm = nullptr;                       // next to the globals, pc_preprocessing_main.h:41-77
static cm_params g_params;

// main(), after ros::init (:511)
int cloudmerge_setup() {
    cm_limits lim = {6, 0, 2u << 20};                // 6 sensors, up to 2 M points per frame
    if (cm_create(&g_cm, 0, &lim) != CM_OK) { ROS_FATAL("no MI355X"); return 1; }
    g_params.leaf[0] = g_params.leaf[1] = g_params.leaf[2] = voxel_size;      // Parameter.h:28
    g_params.min_points_per_voxel = points_per_voxel;                          // Parameter.h:27
    g_params.downsample_all_data = 1;                                          // :174
    g_params.crop_enable = 1;                                                  // getROI :20-40
    g_params.crop_min[0] = -roi_mid;       g_params.crop_max[0] = roi_length - roi_mid;
    g_params.crop_min[1] = -roi_width / 2; g_params.crop_max[1] = roi_width / 2;
    g_params.crop_min[2] = roi_z_min;      g_params.crop_max[2] = roi_z_max;
    g_params.required_sensor_mask = 0b101111;        // fr, fl, rr, rl, livox; top_middle optional (:134-136)
    return 0;
}

// after the six lookupTransform calls succeed (:556-562) — once
void cloudmerge_transforms() {
    const tf::StampedTransform* stf[6] = {&front_right_stf, &front_left_stf, &rear_right_stf,
                                          &rear_left_stf, &top_middle_stf, &front_middle_stf};
    for (int s = 0; s < 6; ++s) {
        tf::Transform t(stf[s]->getRotation(), stf[s]->getOrigin());           // as :320
        const tf::Quaternion q = t.getRotation();                               // what pcl_ros would read
        const double qq[4] = {q.x(), q.y(), q.z(), q.w()};
        const double tt[3] = {t.getOrigin().x(), t.getOrigin().y(), t.getOrigin().z()};
        cm_set_sensor_transform(g_cm, s, qq, tt);
    }
}

// each callback (:318-337 and siblings): subscribe to sensor_msgs::PointCloud2 instead of
// pcl::PointCloud<PointXYZI> and hand the payload over — replaces transformPointCloud + getROI.
// Runs on the AsyncSpinner's threads beside the loop (:513): cm_submit_cloud never waits for a merge.
void callbackFrontRight(const sensor_msgs::PointCloud2::ConstPtr& m) {
    uint32_t ox = 0, oy = 4, oz = 8, oi = CM_NO_FIELD;
    for (const auto& f : m->fields) {
        if (f.name == "x") ox = f.offset; else if (f.name == "y") oy = f.offset;
        else if (f.name == "z") oz = f.offset; else if (f.name == "intensity") oi = f.offset;
    }
    cm_submit_cloud(g_cm, 0 /* sensor slot */, m->data.data(), m->width * m->height, m->point_step, ox, oy, oz, oi);
}

// main loop body (:570-580): replaces fusePointclouds + voxelgrid + the voxel leg of publishPointcloud
void cloudmerge_loop_body() {
    cm_result r;
    if (cm_merge_voxelize(g_cm, &g_params, &r) != CM_NOT_READY && r.status >= 0) {
        sensor_msgs::PointCloud2 voxel_msg;                                    // what pcl::toROSMsg builds (:216)
        voxel_msg.height = r.n_out ? 1 : 0; voxel_msg.width = r.n_out; voxel_msg.point_step = 32;
        voxel_msg.row_step = 32 * r.n_out; voxel_msg.is_dense = true; voxel_msg.is_bigendian = false;
        voxel_msg.fields.resize(4);
        const char* names[4] = {"x", "y", "z", "intensity"}; const uint32_t offs[4] = {0, 4, 8, 16};
        for (int k = 0; k < 4; ++k) { voxel_msg.fields[k].name = names[k]; voxel_msg.fields[k].offset = offs[k];
                                      voxel_msg.fields[k].datatype = sensor_msgs::PointField::FLOAT32; voxel_msg.fields[k].count = 1; }
        voxel_msg.data.resize(32 * r.n_out);
        cm_result_copy(g_cm, voxel_msg.data.data(), r.n_out, 32);
        voxel_msg.header.stamp = ros::Time::now();                              // :217
        voxel_msg.header.frame_id = "base_footprint";                           // :218
        voxelpub.publish(voxel_msg);                                            // :219
    }
}
// [snippet-end]
