"""PyramidBuilder (reference: src/image_processing/pyramid_builder.py:4-48): a pass-through holder of the two
current images, exactly like the reference (the pyramids themselves are built inside the LK operator)."""


class PyramidBuilder(object):
    def __init__(self, win_size, pyramid_levels, cam0_curr_img_msg, cam1_curr_img_msg):
        self.win_size = win_size
        self.pyramid_levels = pyramid_levels
        self.cam0_curr_img_msg = cam0_curr_img_msg
        self.cam1_curr_img_msg = cam1_curr_img_msg
        self.curr_cam0_pyramid = None
        self.curr_cam1_pyramid = None

    def create_image_pyramids(self):
        self.curr_cam0_pyramid = self.cam0_curr_img_msg.image
        self.curr_cam1_pyramid = self.cam1_curr_img_msg.image
        return self.curr_cam0_pyramid, self.curr_cam1_pyramid
